"""Hot-path hyper-parameters of the two reference configs this build targets.

nuScenes: configs/sparsefusion_nusc_voxel_LC_r50.py:1,6,153-186.
Waymo:    configs/transfusion_waymo_voxel_L.py:1-3,109-150.
The dict keys are the reference's own (`pts_voxel_layer`, `pts_voxel_encoder`,
`pts_middle_encoder`, `pts_backbone`, `pts_neck`), so a reference config can be passed through unchanged.
"""

NUSC = dict(
    point_cloud_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0],
    voxel_size=[0.075, 0.075, 0.2],
    out_size_factor=8,
    pts_voxel_layer=dict(
        max_num_points=10,
        voxel_size=[0.075, 0.075, 0.2],
        max_voxels=(120000, 160000),
        point_cloud_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]),
    pts_voxel_encoder=dict(type='HardSimpleVFE', num_features=5),
    pts_middle_encoder=dict(
        type='SparseEncoder',
        in_channels=5,
        sparse_shape=[41, 1440, 1440],
        output_channels=128,
        order=('conv', 'norm', 'act'),
        encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
        encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)),
        block_type='basicblock'),
    pts_backbone=dict(
        type='SECOND',
        in_channels=256,
        out_channels=[128, 256],
        layer_nums=[5, 5],
        layer_strides=[1, 2],
        norm_cfg=dict(type='BN', eps=0.001, momentum=0.01),
        conv_cfg=dict(type='Conv2d', bias=False)),
    pts_neck=dict(
        type='SECONDFPN',
        in_channels=[128, 256],
        out_channels=[256, 256],
        upsample_strides=[1, 2],
        norm_cfg=dict(type='BN', eps=0.001, momentum=0.01),
        upsample_cfg=dict(type='deconv', bias=False),
        use_conv_for_no_stride=True),
)

WAYMO = dict(
    point_cloud_range=[-75.2, -75.2, -2, 75.2, 75.2, 4],
    voxel_size=[0.1, 0.1, 0.15],
    out_size_factor=8,
    pts_voxel_layer=dict(
        max_num_points=5,
        voxel_size=[0.1, 0.1, 0.15],
        max_voxels=150000,
        point_cloud_range=[-75.2, -75.2, -2, 75.2, 75.2, 4]),
    pts_voxel_encoder=dict(
        type='HardVFE',
        in_channels=5,
        feat_channels=[64],
        with_distance=False,
        with_cluster_center=False,
        with_voxel_center=False,
        voxel_size=[0.1, 0.1, 0.15],
        norm_cfg=dict(type='BN1d', eps=0.001, momentum=0.01),
        point_cloud_range=[-75.2, -75.2, -2, 75.2, 75.2, 4]),
    pts_middle_encoder=dict(
        type='SparseEncoder',
        in_channels=64,
        sparse_shape=[41, 1504, 1504],
        output_channels=128,
        order=('conv', 'norm', 'act'),
        encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
        encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)),
        block_type='basicblock'),
    pts_backbone=dict(
        type='SECOND',
        in_channels=256,
        out_channels=[128, 256],
        layer_nums=[5, 5],
        layer_strides=[1, 2],
        norm_cfg=dict(type='BN', eps=0.001, momentum=0.01),
        conv_cfg=dict(type='Conv2d', bias=False)),
    pts_neck=dict(
        type='SECONDFPN',
        in_channels=[128, 256],
        out_channels=[256, 256],
        upsample_strides=[1, 2],
        norm_cfg=dict(type='BN', eps=0.001, momentum=0.01),
        upsample_cfg=dict(type='deconv', bias=False),
        use_conv_for_no_stride=True),
)

# configs/transfusion_nusc_pillar_L.py:1-3,152-184 (the pillar family)
PILLAR = dict(
    point_cloud_range=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0],
    voxel_size=[0.2, 0.2, 8],
    out_size_factor=4,
    pts_voxel_layer=dict(
        max_num_points=20,
        voxel_size=[0.2, 0.2, 8],
        max_voxels=(30000, 60000),
        point_cloud_range=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]),
    pts_voxel_encoder=dict(
        type='PillarFeatureNet',
        in_channels=5,
        feat_channels=[64],
        with_distance=False,
        voxel_size=[0.2, 0.2, 8],
        norm_cfg=dict(type='BN1d', eps=0.001, momentum=0.01),
        point_cloud_range=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]),
    pts_middle_encoder=dict(type='PointPillarsScatter', in_channels=64, output_shape=(512, 512)),
    pts_backbone=dict(
        type='SECOND',
        in_channels=64,
        out_channels=[64, 128, 256],
        layer_nums=[3, 5, 5],
        layer_strides=[2, 2, 2],
        norm_cfg=dict(type='BN', eps=0.001, momentum=0.01),
        conv_cfg=dict(type='Conv2d', bias=False)),
    pts_neck=dict(
        type='SECONDFPN',
        in_channels=[64, 128, 256],
        out_channels=[128, 128, 128],
        upsample_strides=[0.5, 1, 2],
        norm_cfg=dict(type='BN', eps=0.001, momentum=0.01),
        upsample_cfg=dict(type='deconv', bias=False),
        use_conv_for_no_stride=True),
)

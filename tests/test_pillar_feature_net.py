"""PillarFeatureNet / PFNLayer (mmdet3d/models/voxel_encoders/pillar_encoder.py:11-150,
voxel_encoders/utils.py:106-180) and the pillar LiDAR branch of configs/transfusion_nusc_pillar_L.py
(Voxelization -> PillarFeatureNet -> PointPillarsScatter -> SECOND -> SECONDFPN).

The reference holds a shape test only (tests/test_models/test_voxel_encoder/test_voxel_encoders.py:
6-22: in_channels 5, feat_channels [64], output [97297, 64]) => values are "parity unpinned" by
upstream data; the numpy oracle is pinned by a torch transcription of the reference's forward,
statement by statement including the in-place aliasing of legacy mode, and the HIP kernel is
compared with the oracle.  Tolerance: fp32; besides the summation order of the dot products, the
cluster-centre decoration subtracts a mean of up to 20 coordinates of magnitude <= 51.2 m whose sum
rounds at ~6e-5 depending on the order (numpy pairwise / torch and the kernel sequential), and the
offsets it feeds the layer are small: |a-b| <= 1e-5|b| + 1e-5 max|b|."""
import numpy as np
import pytest
import torch
from torch import nn

from helpers import assert_close, to_dev, torch_state_dict
from test_hard_vfe import make_case

PIL = dict(vs=[0.2, 0.2, 8], rng=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0])


def make_pfn_layers(seed, chans):
    rng = np.random.default_rng(seed)
    layers = []
    for i in range(len(chans) - 1):
        last = i == len(chans) - 2
        cin, co = chans[i], (chans[i + 1] if last else chans[i + 1] // 2)
        w = rng.uniform(-1, 1, (co, cin)).astype(np.float32) / np.sqrt(cin)
        bn = dict(weight=rng.uniform(0.5, 1.5, co), bias=rng.normal(0, 0.3, co),
                  running_mean=rng.normal(0, 0.3, co), running_var=rng.uniform(0.5, 1.5, co))
        layers.append((w, {k: v.astype(np.float32) for k, v in bn.items()}))
    return layers


def torch_transcription(feats, num, coors, layers, voxel_size, pc_range, cluster, center, dist,
                        legacy):
    """pillar_encoder.py:103-150 + utils.py:159-180 with the reference's own module types, eval."""
    features = torch.from_numpy(feats.copy())
    num_points = torch.from_numpy(num)
    c = torch.from_numpy(coors)
    vx, vy = voxel_size[0], voxel_size[1]
    x_offset, y_offset = vx / 2 + pc_range[0], vy / 2 + pc_range[1]
    features_ls = [features]
    if cluster:
        points_mean = features[:, :, :3].sum(dim=1, keepdim=True) / \
            num_points.type_as(features).view(-1, 1, 1)
        features_ls.append(features[:, :, :3] - points_mean)
    dtype = features.dtype
    if center:
        if not legacy:
            f_center = torch.zeros_like(features[:, :, :2])
            f_center[:, :, 0] = features[:, :, 0] - (c[:, 3].to(dtype).unsqueeze(1) * vx + x_offset)
            f_center[:, :, 1] = features[:, :, 1] - (c[:, 2].to(dtype).unsqueeze(1) * vy + y_offset)
        else:
            f_center = features[:, :, :2]
            f_center[:, :, 0] = f_center[:, :, 0] - (
                c[:, 3].type_as(features).unsqueeze(1) * vx + x_offset)
            f_center[:, :, 1] = f_center[:, :, 1] - (
                c[:, 2].type_as(features).unsqueeze(1) * vy + y_offset)
        features_ls.append(f_center)
    if dist:
        features_ls.append(torch.norm(features[:, :, :3], 2, 2, keepdim=True))
    features = torch.cat(features_ls, dim=-1)
    voxel_count = features.shape[1]
    mask = num_points.unsqueeze(1).int() > torch.arange(voxel_count, dtype=torch.int).view(1, -1)
    features = features * mask.unsqueeze(-1).type_as(features)
    for i, (w, bn) in enumerate(layers):
        lin = nn.Linear(w.shape[1], w.shape[0], bias=False)
        norm = nn.BatchNorm1d(w.shape[0], eps=1e-3, momentum=0.01).eval()
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(w))
            for k in ('weight', 'bias', 'running_mean', 'running_var'):
                getattr(norm, k).copy_(torch.from_numpy(bn[k]))
            x = lin(features)
            x = norm(x.permute(0, 2, 1).contiguous()).permute(0, 2, 1).contiguous()
            x = torch.relu(x)
            x_max = torch.max(x, dim=1, keepdim=True)[0]
            if i == len(layers) - 1:
                features = x_max
            else:
                features = torch.cat([x, x_max.repeat(1, features.shape[1], 1)], dim=2)
    return features.squeeze().numpy()


CASES = [  # (feat_channels, cluster, center, dist, legacy, max_points, in_channels)
    ([64], True, True, False, True, 20, 5),      # configs/transfusion_nusc_pillar_L.py:157-165
    ([64], True, True, False, False, 20, 5),
    ([32, 64], True, True, True, True, 8, 4),    # two PFN layers, distance sees the overwritten x / y
    ([16], False, False, True, True, 6, 4),
]


def _cin(c, cluster, center, dist):
    return c + 3 * cluster + 2 * center + dist


@pytest.mark.parametrize('chans,cluster,center,dist,legacy,p,c', CASES)
def test_oracle_matches_torch_transcription(oracle, chans, cluster, center, dist, legacy, p, c):
    feats, num, coors = make_case(15, 600, p, c, PIL['vs'], PIL['rng'])
    layers = make_pfn_layers(16, [_cin(c, cluster, center, dist)] + chans)
    a = oracle.pillar_feature_net(feats, num, coors, layers, PIL['vs'], PIL['rng'], 1e-3, cluster,
                                  center, dist, legacy)
    b = torch_transcription(feats, num, coors, layers, PIL['vs'], PIL['rng'], cluster, center, dist,
                            legacy)
    assert a.shape == b.shape == (600, chans[-1])
    assert_close(a, b, 1e-5, 1e-5)
    assert np.abs(b).max() > 1e-2


def test_construction_matches_reference():
    """tests/test_models/test_voxel_encoder/test_voxel_encoders.py:6-22 (constructor part)."""
    from sparsefusion_amd.registry import VOXEL_ENCODERS
    from sparsefusion_amd.voxel_encoder import PFNLayer
    pfn = VOXEL_ENCODERS.build(dict(type='PillarFeatureNet', in_channels=5, feat_channels=[64],
                                    with_distance=False, voxel_size=(0.2, 0.2, 8),
                                    point_cloud_range=(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0),
                                    norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01)))
    assert pfn.in_channels == 10 and pfn.legacy
    assert isinstance(pfn.pfn_layers[0], PFNLayer) and pfn.pfn_layers[0].last_vfe
    assert pfn.pfn_layers[0].linear.weight.shape == (64, 10) and pfn.pfn_layers[0].units == 64
    two = VOXEL_ENCODERS.build(dict(type='PillarFeatureNet', in_channels=4, feat_channels=[32, 64]))
    assert two.pfn_layers[0].units == 16 and not two.pfn_layers[0].last_vfe
    assert two.pfn_layers[1].linear.weight.shape == (64, 32)
    assert set(two.state_dict()) >= {'pfn_layers.0.linear.weight', 'pfn_layers.1.norm.running_var'}
    with pytest.raises(NotImplementedError):
        VOXEL_ENCODERS.build(dict(type='PillarFeatureNet', mode='avg'))


def _module(dev, chans, cluster, center, dist, legacy, c, layers):
    from sparsefusion_amd.voxel_encoder import PillarFeatureNet
    pfn = PillarFeatureNet(in_channels=c, feat_channels=chans, with_distance=dist,
                           with_cluster_center=cluster, with_voxel_center=center,
                           voxel_size=PIL['vs'], point_cloud_range=PIL['rng'], legacy=legacy)
    sd = {}
    for i, (w, bn) in enumerate(layers):
        sd[f'pfn_layers.{i}.linear.weight'] = torch.from_numpy(w)
        for k, v in bn.items():
            sd[f'pfn_layers.{i}.norm.{k}'] = torch.from_numpy(v)
        sd[f'pfn_layers.{i}.norm.num_batches_tracked'] = torch.tensor(0)
    pfn.load_state_dict(sd)
    return pfn.to(dev).eval()


@pytest.mark.gpu
@pytest.mark.parametrize('chans,cluster,center,dist,legacy,p,c', CASES)
@pytest.mark.parametrize('m', [2, 4000])
def test_pillar_feature_net_gpu_matches_oracle(dev, oracle, chans, cluster, center, dist, legacy, p,
                                               c, m):
    feats, num, coors = make_case(17 + m, m, p, c, PIL['vs'], PIL['rng'])
    layers = make_pfn_layers(18, [_cin(c, cluster, center, dist)] + chans)
    pfn = _module(dev, chans, cluster, center, dist, legacy, c, layers)
    d_feats = to_dev(feats, dev)
    with torch.no_grad():
        got = pfn(d_feats, to_dev(num, dev), to_dev(coors, dev))
    torch.cuda.synchronize()
    assert got.shape == (m, chans[-1])
    expect = oracle.pillar_feature_net(feats, num, coors, layers, PIL['vs'], PIL['rng'], 1e-3,
                                       cluster, center, dist, legacy)
    assert_close(got.cpu().numpy(), expect, 1e-5, 1e-5)
    assert np.array_equal(d_feats.cpu().numpy(), feats)       # the caller's tensor is not modified


@pytest.mark.gpu
def test_pillar_lidar_branch(dev, oracle):
    """configs/transfusion_nusc_pillar_L.py:152-184 end to end on the device: hard voxelization
    (pillars: one voxel in z) -> PillarFeatureNet -> PointPillarsScatter -> SECOND -> SECONDFPN, two
    samples, against the oracle chain.  Pillar canvas 128 x 128 (a quarter-range crop of the config's
    512 x 512) so that the CPU side stays in seconds; channel counts are the config's."""
    from sparsefusion_amd import configs, synth
    from sparsefusion_amd.pillar_scatter import PointPillarsScatter
    from sparsefusion_amd.second import SECOND, SECONDFPN
    from sparsefusion_amd.voxel_encoder import PillarFeatureNet
    from sparsefusion_amd.voxelize import Voxelization
    rng_ = [-12.8, -12.8, -5.0, 12.8, 12.8, 3.0]
    vl = dict(configs.PILLAR['pts_voxel_layer'], point_cloud_range=rng_, max_voxels=(3000, 6000))
    ecfg = dict(configs.PILLAR['pts_voxel_encoder'], point_cloud_range=rng_)
    ecfg.pop('type')
    bcfg, ncfg = dict(configs.PILLAR['pts_backbone']), dict(configs.PILLAR['pts_neck'])
    bcfg.pop('type'), ncfg.pop('type')
    vox = Voxelization(**vl).eval()
    pfn = PillarFeatureNet(**ecfg)
    psd = synth.pillar_feature_net_state_dict(3, pfn.in_channels, ecfg['feat_channels'])
    pfn.load_state_dict(torch_state_dict(psd))
    second, neck = SECOND(**bcfg), SECONDFPN(**ncfg)
    bsd, nsd = synth.second_state_dict(4, **bcfg), synth.second_fpn_state_dict(5, **ncfg)
    second.load_state_dict(torch_state_dict(bsd))
    neck.load_state_dict(torch_state_dict(nsd))
    pfn, second, neck = pfn.to(dev).eval(), second.to(dev).eval(), neck.to(dev).eval()
    scatter = PointPillarsScatter(64, (128, 128))
    frames = [synth.small_frame(70 + b, 9000 + 2000 * b, rng_, spread=6.0, nfeat=5) for b in range(2)]
    with torch.no_grad():
        vs, cs, ns = zip(*[vox(to_dev(p, dev)) for p in frames])
        coors4 = torch.cat([torch.nn.functional.pad(c, (1, 0), value=i) for i, c in enumerate(cs)])
        pillars = pfn(torch.cat(vs), torch.cat(ns), coors4)
        canvas = scatter(pillars, coors4, 2)
        out = neck(second(canvas))[0]
    torch.cuda.synchronize()
    # oracle chain
    o = [oracle.hard_voxelize(p, vl['voxel_size'], rng_, vl['max_num_points'], 6000) for p in frames]
    o_coors = oracle.batch_coors([c for _, c, _ in o])
    assert np.array_equal(coors4.cpu().numpy(), o_coors)
    layers = [(psd['pfn_layers.0.linear.weight'],
               {k: psd[f'pfn_layers.0.norm.{k}'] for k in ('weight', 'bias', 'running_mean',
                                                            'running_var')})]
    o_pil = oracle.pillar_feature_net(np.concatenate([v for v, _, _ in o]),
                                      np.concatenate([n for _, _, n in o]), o_coors, layers,
                                      ecfg['voxel_size'], rng_, 1e-3)
    assert_close(pillars.cpu().numpy(), o_pil, 1e-5, 1e-5)
    o_canvas = oracle.pillar_scatter(o_pil, o_coors, 2, 128, 128)
    o_out = oracle.second_fpn_forward(nsd, oracle.second_forward(bsd, o_canvas, bcfg['layer_nums'],
                                                                 bcfg['layer_strides']),
                                      ncfg['upsample_strides'], True)[0]
    assert tuple(out.shape) == o_out.shape == (2, 384, 32, 32)
    assert_close(out.cpu().numpy(), o_out, rtol=2e-4, atol_scale=2e-5)

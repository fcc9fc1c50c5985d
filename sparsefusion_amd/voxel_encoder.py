"""`HardSimpleVFE`: drop-in for mmdet3d/models/voxel_encoders/voxel_encoder.py:12-44."""
import torch
from torch import nn

from . import _lib
from .registry import VOXEL_ENCODERS


@VOXEL_ENCODERS.register_module()
class HardSimpleVFE(nn.Module):
    """Mean of the points in each voxel (the SECOND voxel feature encoder)."""

    def __init__(self, num_features=4):
        super().__init__()
        self.num_features = num_features
        self.fp16_enabled = False

    def forward(self, features, num_points, coors=None):
        """features (M, max_points, F) f32, num_points (M,) -> (M, num_features) contiguous.
        Sum over every slot (zero padded ones included) divided by num_points, as the
        reference's `features[:, :, :nf].sum(dim=1) / num_points` (voxel_encoder.py:42-44)."""
        _lib.require(features, torch.float32, 'features')
        m, mp, f = features.shape
        out = torch.empty((m, self.num_features), dtype=torch.float32, device=features.device)
        _lib.check(_lib.lib().sf_voxel_mean(
            _lib.ptr(features.contiguous()), _lib.ptr(num_points.int().contiguous()), None, m, mp,
            f, self.num_features, out.data_ptr(), self.num_features,
            _lib.stream(features.device)))
        return out


def decorated_channels(cin_raw, decorate):
    """Input width of sf_vfe_layer's Linear for `cin_raw` raw channels and the decorate bits
    (include/sparsefusion_hip.h: 1 cluster centre +3, 2 voxel centre +3 (+2 with bit 16), 4
    distance +1)."""
    return cin_raw + 3 * (decorate & 1) + ((2 if decorate & 16 else 3) if decorate & 2 else 0) + \
        (1 if decorate & 4 else 0)


class VFELayer(nn.Module):
    """mmdet3d/models/voxel_encoders/utils.py:31-96: Linear(bias=False) -> BN1d -> ReLU -> max over
    the points of a voxel [-> concat].  Same attributes / state-dict keys (`linear`, `norm`)."""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01),
                 max_out=True, cat_max=True):
        super().__init__()
        from .registry import build_norm_layer
        self.fp16_enabled = False
        self.cat_max = cat_max
        self.max_out = max_out
        self.norm = build_norm_layer(norm_cfg, out_channels)[1]
        self.linear = nn.Linear(in_channels, out_channels, bias=False)

    def _run(self, inputs, decorate=0, num_points=None, coors=None, voxel_size=None, pc_range=None):
        from . import spconv
        _lib.require(inputs, torch.float32, 'inputs')
        if self.norm.training:
            raise NotImplementedError('VFELayer runs eval-mode BatchNorm only (the LiDAR branch '
                                      'is frozen in the target configs)')
        m, p, c = inputs.shape
        cout = self.linear.out_features
        # the kernel derives its input width from the raw channels and the decorate bits and
        # indexes the (cout, cin) weight with it: a mismatch must fail here, as nn.Linear would
        cin = decorated_channels(c, decorate)
        if cin != self.linear.in_features:
            raise RuntimeError(f'VFE layer expects {self.linear.in_features} input channels, got '
                               f'{c} raw + {cin - c} decoration channels = {cin}')
        scale, shift = spconv.fold_bn(self.norm)
        mode = 1 if not self.max_out else (2 if self.cat_max else 0)
        shape = {0: (m, cout), 1: (m, p, cout), 2: (m, p, 2 * cout)}[mode]
        out = torch.empty(shape, dtype=torch.float32, device=inputs.device)
        _lib.check(_lib.lib().sf_vfe_layer(
            _lib.ptr(inputs.contiguous()), c, p,
            _lib.ptr(num_points.int().contiguous()) if num_points is not None else None,
            _lib.ptr(coors.int().contiguous()) if coors is not None else None, None, m, decorate,
            _lib.f3(voxel_size) if voxel_size is not None else None,
            _lib.f6(pc_range) if pc_range is not None else None,
            _lib.ptr(self.linear.weight.detach().float().contiguous()), cout, _lib.ptr(scale),
            _lib.ptr(shift), mode, out.data_ptr(), _lib.stream(inputs.device)))
        return _lib.forward_only(out, inputs, self.linear.weight, self.norm.weight, self.norm.bias)

    def forward(self, inputs):
        return self._run(inputs)


@VOXEL_ENCODERS.register_module()
class HardVFE(nn.Module):
    """Drop-in for mmdet3d/models/voxel_encoders/voxel_encoder.py:283-450 (the voxel encoder of
    configs/transfusion_waymo_voxel_L.py:114-124).  The decorations, the padding mask and the
    first VFELayer run as ONE kernel; image fusion (`fusion_layer`) belongs to the camera branch
    and is not built."""

    def __init__(self, in_channels=4, feat_channels=[], with_distance=False,
                 with_cluster_center=False, with_voxel_center=False, voxel_size=(0.2, 0.2, 4),
                 point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), mode='max', fusion_layer=None,
                 return_point_feats=False):
        super().__init__()
        assert len(feat_channels) > 0
        if fusion_layer is not None:
            raise NotImplementedError('point/image fusion layers are outside the LiDAR hot path')
        if with_cluster_center:
            in_channels += 3
        if with_voxel_center:
            in_channels += 3
        if with_distance:
            in_channels += 3      # sic: the reference reserves 3 channels for ONE distance column
        self.in_channels = in_channels
        self._with_distance = with_distance
        self._with_cluster_center = with_cluster_center
        self._with_voxel_center = with_voxel_center
        self.return_point_feats = return_point_feats
        self.fp16_enabled = False
        self.vx, self.vy, self.vz = voxel_size[0], voxel_size[1], voxel_size[2]
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.z_offset = self.vz / 2 + point_cloud_range[2]
        self.voxel_size = list(voxel_size)
        self.point_cloud_range = point_cloud_range
        from .scatter_points import DynamicScatter
        self.scatter = DynamicScatter(voxel_size, point_cloud_range, True)
        chans = [self.in_channels] + list(feat_channels)
        layers = []
        for i in range(len(chans) - 1):
            cin, cout = chans[i], chans[i + 1]
            if i > 0:
                cin *= 2
            last = i == len(chans) - 2
            layers.append(VFELayer(cin, cout, norm_cfg=norm_cfg, max_out=True, cat_max=not last))
        self.vfe_layers = nn.ModuleList(layers)
        self.num_vfe = len(layers)
        self.fusion_layer = None

    def forward(self, features, num_points, coors, img_feats=None, img_metas=None):
        """features (M, P, C) f32, num_points (M,), coors (M, 4) (b, z, y, x) -> (M, feat_channels[-1])."""
        if self._with_distance:
            # the reference's own forward fails here: it builds Linear(C+3, ...) but concatenates
            # one distance column (voxel_encoder.py:337-338 vs :430-432)
            raise NotImplementedError('with_distance: the reference allots 3 channels to a '
                                      '1-channel decoration and cannot run this configuration')
        deco = (1 if self._with_cluster_center else 0) | (2 if self._with_voxel_center else 0) | 8
        x = features
        for i, vfe in enumerate(self.vfe_layers):
            if i == 0:
                x = vfe._run(x, deco, num_points, coors, self.voxel_size,
                             list(self.point_cloud_range))
            else:
                x = vfe(x)
        return x


class PFNLayer(VFELayer):
    """mmdet3d/models/voxel_encoders/utils.py:106-180: Linear(bias=False) -> BN1d -> ReLU -> max over
    the points of a pillar; every layer but the last has out_channels // 2 units and concatenates
    the repeated max.  Same attributes / state-dict keys (`linear`, `norm`, `units`, `last_vfe`)."""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01),
                 last_layer=False, mode='max'):
        assert mode in ['max', 'avg']
        if mode != 'max':
            raise NotImplementedError("PFNLayer mode 'avg' (the target configs use 'max')")
        if not last_layer:
            out_channels = out_channels // 2
        super().__init__(in_channels, out_channels, norm_cfg=norm_cfg, max_out=True,
                         cat_max=not last_layer)
        self.name = 'PFNLayer'
        self.last_vfe = last_layer
        self.units = out_channels
        self.mode = mode

    def forward(self, inputs, num_voxels=None, aligned_distance=None):
        if aligned_distance is not None:
            raise NotImplementedError('aligned_distance is not used by PillarFeatureNet')
        out = self._run(inputs)
        return out.unsqueeze(1) if self.last_vfe else out     # (N, 1, C) as the reference's x_max


@VOXEL_ENCODERS.register_module()
class PillarFeatureNet(nn.Module):
    """Drop-in for mmdet3d/models/voxel_encoders/pillar_encoder.py:11-150 (the voxel encoder of
    configs/transfusion_nusc_pillar_L.py:157-165).  Decorations, padding mask and the first
    PFNLayer run as ONE kernel (sf_vfe_layer with the pillar decorate bits)."""

    def __init__(self, in_channels=4, feat_channels=(64, ), with_distance=False,
                 with_cluster_center=True, with_voxel_center=True, voxel_size=(0.2, 0.2, 4),
                 point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), mode='max', legacy=True):
        super().__init__()
        assert len(feat_channels) > 0
        self.legacy = legacy
        if with_cluster_center:
            in_channels += 3
        if with_voxel_center:
            in_channels += 2
        if with_distance:
            in_channels += 1
        self._with_distance = with_distance
        self._with_cluster_center = with_cluster_center
        self._with_voxel_center = with_voxel_center
        self.fp16_enabled = False
        self.in_channels = in_channels
        chans = [in_channels] + list(feat_channels)
        self.pfn_layers = nn.ModuleList([
            PFNLayer(chans[i], chans[i + 1], norm_cfg=norm_cfg, last_layer=i == len(chans) - 2,
                     mode=mode) for i in range(len(chans) - 1)])
        self.vx = voxel_size[0]
        self.vy = voxel_size[1]
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.voxel_size = list(voxel_size)
        self.point_cloud_range = point_cloud_range

    def forward(self, features, num_points, coors):
        """features (N, M, C) f32, num_points (N,), coors (N, 4) (b, z, y, x) -> (N, feat_channels[-1])
        (`.squeeze()`d like the reference's return value).  Unlike the reference's legacy mode the
        caller's `features` tensor is left untouched (the in-place overwrite of its x / y columns,
        pillar_encoder.py:124-130, happens on the kernel's staged copy)."""
        deco = 8 | (1 if self._with_cluster_center else 0) | (4 if self._with_distance else 0)
        if self._with_voxel_center:
            deco |= 2 | 16 | (32 if self.legacy else 0)
        x = features
        for i, pfn in enumerate(self.pfn_layers):
            if i == 0:
                x = pfn._run(x, deco, num_points, coors, self.voxel_size,
                             list(self.point_cloud_range))
            else:
                x = pfn._run(x)
        return x.squeeze()

"""HardVFE / VFELayer (mmdet3d/models/voxel_encoders/voxel_encoder.py:283-450, utils.py:31-96).

The reference holds only a shape test for it (tests/test_models/test_voxel_encoder/
test_voxel_encoders.py), so the numpy oracle is pinned by a torch transcription of the reference's
forward (nn.Linear / nn.BatchNorm1d modules, the reference's own statements) on the CPU --
"parity unpinned" by upstream data -- and the HIP kernel is compared with the oracle on the GPU.
Tolerance: fp32, different summation order in the C-channel dot products:
|a-b| <= 1e-5|b| + 1e-6 max|b|."""
import numpy as np
import pytest
import torch
from torch import nn

from helpers import assert_close


def make_case(seed, m, p, c, voxel_size, pc_range):
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(pc_range[:3], np.float32), np.asarray(pc_range[3:], np.float32)
    vs = np.asarray(voxel_size, np.float32)
    grid = np.round((hi - lo) / vs).astype(np.int64)
    cell = np.stack([rng.integers(0, grid[i], m) for i in range(3)], 1)       # x, y, z cell
    num = rng.integers(1, p + 1, m).astype(np.int32)
    feats = np.zeros((m, p, c), np.float32)
    xyz = lo + (cell[:, None, :] + rng.random((m, p, 3))) * vs
    feats[:, :, :3] = xyz
    feats[:, :, 3:] = rng.normal(0, 1, (m, p, c - 3))
    feats *= (num[:, None] > np.arange(p)[None, :])[:, :, None]               # zero padded slots
    coors = np.stack([rng.integers(0, 2, m), cell[:, 2], cell[:, 1], cell[:, 0]], 1).astype(np.int32)
    return feats.astype(np.float32), num, coors


def make_layers(seed, chans):
    rng = np.random.default_rng(seed)
    layers = []
    for i in range(len(chans) - 1):
        cin = chans[i] * (2 if i > 0 else 1)
        co = chans[i + 1]
        w = rng.uniform(-1, 1, (co, cin)).astype(np.float32) / np.sqrt(cin)
        bn = dict(weight=rng.uniform(0.5, 1.5, co), bias=rng.normal(0, 0.3, co),
                  running_mean=rng.normal(0, 0.3, co), running_var=rng.uniform(0.5, 1.5, co))
        layers.append((w, {k: v.astype(np.float32) for k, v in bn.items()}))
    return layers


def torch_transcription(feats, num, coors, layers, voxel_size, pc_range, cluster, center):
    """The reference's statements with its own module types (voxel_encoder.py:404-446,
    utils.py:82-104), eval mode."""
    features = torch.from_numpy(feats)
    num_points = torch.from_numpy(num)
    c = torch.from_numpy(coors)
    vx, vy, vz = voxel_size
    ls = [features]
    if cluster:
        points_mean = features[:, :, :3].sum(dim=1, keepdim=True) / \
            num_points.type_as(features).view(-1, 1, 1)
        ls.append(features[:, :, :3] - points_mean)
    if center:
        f_center = features.new_zeros(size=(features.size(0), features.size(1), 3))
        f_center[:, :, 0] = features[:, :, 0] - (
            c[:, 3].type_as(features).unsqueeze(1) * vx + (vx / 2 + pc_range[0]))
        f_center[:, :, 1] = features[:, :, 1] - (
            c[:, 2].type_as(features).unsqueeze(1) * vy + (vy / 2 + pc_range[1]))
        f_center[:, :, 2] = features[:, :, 2] - (
            c[:, 1].type_as(features).unsqueeze(1) * vz + (vz / 2 + pc_range[2]))
        ls.append(f_center)
    x = torch.cat(ls, dim=-1)
    mask = num_points.unsqueeze(1).int() > torch.arange(x.shape[1], dtype=torch.int).view(1, -1)
    x = x * mask.unsqueeze(-1).type_as(x)
    for i, (w, bn) in enumerate(layers):
        lin = nn.Linear(w.shape[1], w.shape[0], bias=False)
        norm = nn.BatchNorm1d(w.shape[0], eps=1e-3, momentum=0.01).eval()
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(w))
            norm.weight.copy_(torch.from_numpy(bn['weight']))
            norm.bias.copy_(torch.from_numpy(bn['bias']))
            norm.running_mean.copy_(torch.from_numpy(bn['running_mean']))
            norm.running_var.copy_(torch.from_numpy(bn['running_var']))
            y = lin(x)
            y = norm(y.permute(0, 2, 1).contiguous()).permute(0, 2, 1).contiguous()
            pointwise = torch.relu(y)
            aggregated = torch.max(pointwise, dim=1, keepdim=True)[0]
            if i == len(layers) - 1:
                x = aggregated.squeeze(1)
            else:
                x = torch.cat([pointwise, aggregated.repeat(1, x.shape[1], 1)], dim=2)
    return x.numpy()


WAYMO = dict(vs=[0.1, 0.1, 0.15], rng=[-75.2, -75.2, -2, 75.2, 75.2, 4])
CASES = [  # (feat_channels, cluster, center, max_points, in_channels)
    ([64], False, False, 5, 5),          # configs/transfusion_waymo_voxel_L.py:114-124
    ([32, 128], True, True, 10, 4),      # the DV-SECOND style two-layer encoder
    ([16], True, False, 3, 5),
]


@pytest.mark.parametrize('chans,cluster,center,p,c', CASES)
def test_oracle_matches_torch_transcription(oracle, chans, cluster, center, p, c):
    feats, num, coors = make_case(5, 700, p, c, WAYMO['vs'], WAYMO['rng'])
    cin = c + 3 * cluster + 3 * center
    layers = make_layers(6, [cin] + chans)
    a = oracle.hard_vfe(feats, num, coors, layers, WAYMO['vs'], WAYMO['rng'], 1e-3, cluster, center)
    b = torch_transcription(feats, num, coors, layers, WAYMO['vs'], WAYMO['rng'], cluster, center)
    assert a.shape == b.shape == (700, chans[-1])
    assert_close(a, b, 1e-5, 1e-6)
    assert np.abs(b).max() > 1e-2


def _module(dev, chans, cluster, center, c, layers):
    from sparsefusion_amd.voxel_encoder import HardVFE
    vfe = HardVFE(in_channels=c, feat_channels=chans, with_cluster_center=cluster,
                  with_voxel_center=center, voxel_size=WAYMO['vs'], point_cloud_range=WAYMO['rng'],
                  norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01))
    sd = {}
    for i, (w, bn) in enumerate(layers):
        sd[f'vfe_layers.{i}.linear.weight'] = torch.from_numpy(w)
        for k, v in bn.items():
            sd[f'vfe_layers.{i}.norm.{k}'] = torch.from_numpy(v)
        sd[f'vfe_layers.{i}.norm.num_batches_tracked'] = torch.tensor(0)
    vfe.load_state_dict(sd)            # the reference's key names
    return vfe.to(dev).eval()


@pytest.mark.gpu
@pytest.mark.parametrize('chans,cluster,center,p,c', CASES)
@pytest.mark.parametrize('m', [0, 1, 5000])
def test_hard_vfe_gpu_matches_oracle(dev, oracle, chans, cluster, center, p, c, m):
    feats, num, coors = make_case(7 + m, max(m, 1), p, c, WAYMO['vs'], WAYMO['rng'])
    feats, num, coors = feats[:m], num[:m], coors[:m]
    cin = c + 3 * cluster + 3 * center
    layers = make_layers(8, [cin] + chans)
    vfe = _module(dev, chans, cluster, center, c, layers)
    with torch.no_grad():
        got = vfe(torch.from_numpy(feats).to(dev), torch.from_numpy(num).to(dev),
                  torch.from_numpy(coors).to(dev))
    torch.cuda.synchronize()
    assert got.shape == (m, chans[-1])
    if m:
        expect = oracle.hard_vfe(feats, num, coors, layers, WAYMO['vs'], WAYMO['rng'], 1e-3, cluster,
                                 center)
        assert_close(got.cpu().numpy(), expect, 1e-5, 1e-6)


def test_hard_vfe_construction_matches_reference():
    from sparsefusion_amd.voxel_encoder import HardVFE, VFELayer
    vfe = HardVFE(in_channels=5, feat_channels=[64], voxel_size=WAYMO['vs'],
                  point_cloud_range=WAYMO['rng'])
    assert vfe.num_vfe == 1 and isinstance(vfe.vfe_layers[0], VFELayer)
    assert vfe.vfe_layers[0].linear.weight.shape == (64, 5) and vfe.vfe_layers[0].linear.bias is None
    assert vfe.vfe_layers[0].max_out and not vfe.vfe_layers[0].cat_max
    assert vfe.vfe_layers[0].norm.eps == 1e-3
    two = HardVFE(in_channels=4, feat_channels=[32, 128], with_cluster_center=True,
                  with_voxel_center=True)
    assert two.in_channels == 10
    assert two.vfe_layers[0].linear.weight.shape == (32, 10) and two.vfe_layers[0].cat_max
    assert two.vfe_layers[1].linear.weight.shape == (128, 64) and not two.vfe_layers[1].cat_max
    with pytest.raises(NotImplementedError):
        HardVFE(in_channels=4, feat_channels=[8], fusion_layer=dict(type='PointFusion'))
    with pytest.raises(RuntimeError):      # no CPU path
        vfe.eval()(torch.zeros(2, 5, 5), torch.ones(2, dtype=torch.int32), torch.zeros(2, 4))

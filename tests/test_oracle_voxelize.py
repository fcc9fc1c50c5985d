"""CPU oracle pinning, voxelization part (runs without a GPU).

1. the reference's own known-answer test (tests/test_models/test_voxel_encoder/
   test_voxel_generator.py:6-22 upstream);
2. the reference's CPU voxelizer compiled from its own sources (oracle/_ref, built by
   oracle/build_ref.py when /root/reference is present) on seeded frames and edge cases;
3. the committed golden fixtures (tests/golden/voxelize_*.npz, generated from oracle/_ref by
   tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from helpers import SMALL_RANGE, SMALL_VOXEL, small_points
from sparsefusion_amd import configs, synth

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_reference_known_answer(oracle):
    np.random.seed(0)
    pts = np.random.rand(1000, 4)
    v, c, n = oracle.hard_voxelize(pts, [0.5, 0.5, 0.5], [0, -40, -3, 70.4, 40, 1], 1000, 20000)
    assert c.tolist() == [[7, 81, 1], [6, 81, 0], [7, 80, 1], [6, 81, 1], [7, 81, 0], [6, 80, 1],
                          [7, 80, 0], [6, 80, 0]]
    assert n.tolist() == [120, 121, 127, 134, 115, 127, 125, 131]
    assert v.shape == (8, 1000, 4)


def _ref():
    from oracle import build_ref
    mod = build_ref.load_prebuilt()
    if mod is None:
        pytest.skip('oracle/_ref not built (reference sources absent)')
    return mod


def _ref_hard(mod, pts, vs, rng, mp, mv):
    import torch
    p = torch.from_numpy(np.ascontiguousarray(pts, np.float32))
    vox = torch.zeros(mv, mp, p.shape[1])
    co = torch.zeros(mv, 3, dtype=torch.int32)
    npv = torch.zeros(mv, dtype=torch.int32)
    k = mod.hard_voxelize(p, vox, co, npv, [float(x) for x in vs], [float(x) for x in rng], mp, mv, 3)
    return vox[:k].numpy(), co[:k].numpy(), npv[:k].numpy()


@pytest.mark.parametrize('n,mp,mv', [(3000, 10, 20000), (4096, 3, 512), (5000, 1, 100),
                                     (2000, 5, 1), (777, 35, 20000)])
def test_oracle_equals_compiled_reference(oracle, n, mp, mv):
    mod = _ref()
    pts = small_points(n, n)
    pts[5, 0] = 8.0        # on the upper bound
    pts[6, 1] = -8.0       # on the lower bound
    pts[7, 2] = 1e30       # far outside (finite)
    ev, ec, en = _ref_hard(mod, pts, SMALL_VOXEL, SMALL_RANGE, mp, mv)
    v, c, k = oracle.hard_voxelize(pts, SMALL_VOXEL, SMALL_RANGE, mp, mv)
    assert np.array_equal(c, ec) and np.array_equal(k, en) and np.array_equal(v, ev)


def test_oracle_equals_compiled_reference_full_frame(oracle):
    import torch
    mod = _ref()
    cfg = configs.NUSC['pts_voxel_layer']
    pts = synth.nuscenes_frame(1)
    ev, ec, en = _ref_hard(mod, pts, cfg['voxel_size'], cfg['point_cloud_range'], 10, 160000)
    v, c, k = oracle.hard_voxelize(pts, cfg['voxel_size'], cfg['point_cloud_range'], 10, 160000)
    assert np.array_equal(c, ec) and np.array_equal(k, en) and np.array_equal(v, ev)
    dyn = torch.zeros(pts.shape[0], 3, dtype=torch.int32)
    mod.dynamic_voxelize(torch.from_numpy(pts), dyn, [float(x) for x in cfg['voxel_size']],
                         [float(x) for x in cfg['point_cloud_range']], 3)
    assert np.array_equal(oracle.dynamic_voxelize(pts, cfg['voxel_size'], cfg['point_cloud_range']),
                          dyn.numpy())


def test_nan_inf_rejected(oracle):
    # the reference reaches "rejected" through an out-of-range float->int cast (UB, INT_MIN on
    # x86-64); the oracle states that outcome explicitly
    pts = small_points(1, 50)
    pts[0, 0], pts[1, 1], pts[2, 2] = np.nan, np.inf, -np.inf
    c = oracle.dynamic_voxelize(pts, SMALL_VOXEL, SMALL_RANGE)
    assert (c[:3] == -1).all() and (c[3:] >= 0).all()


@pytest.mark.parametrize('name', ['voxelize_kitti_like', 'voxelize_small_caps'])
def test_golden_voxelize(oracle, name):
    g = np.load(os.path.join(GOLD, name + '.npz'))
    v, c, n = oracle.hard_voxelize(g['points'], g['voxel_size'], g['coors_range'],
                                   int(g['max_points']), int(g['max_voxels']))
    assert np.array_equal(c, g['coors']) and np.array_equal(n, g['num_points'])
    assert np.array_equal(v, g['voxels'])


def test_vfe_mean_matches_torch_statement(oracle):
    import torch
    pts = small_points(3, 4000)
    v, c, n = oracle.hard_voxelize(pts, SMALL_VOXEL, SMALL_RANGE, 10, 20000)
    tv, tn = torch.from_numpy(v), torch.from_numpy(n)
    ref = tv[:, :, :5].sum(dim=1, keepdim=False) / tn.type_as(tv).view(-1, 1)  # voxel_encoder.py:42-43
    assert np.allclose(oracle.vfe_mean(v, n, 5), ref.numpy(), rtol=1e-6, atol=1e-6)

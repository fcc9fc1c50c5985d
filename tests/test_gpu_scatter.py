"""DynamicScatter on the GPU: the reference's own test recipe
(tests/test_models/test_voxel_encoder/test_dynamic_scatter.py:12-51) and parity with the oracle.
Coordinates, point->voxel map, counts and max are exact; sums / means use float atomics (as the
reference's do, scatter_points.py:60-61 quotes 5e-7) -> rtol 1e-5, atol 2e-4 on values in +-50."""
import numpy as np
import pytest
import torch

from sparsefusion_amd import configs, synth
from sparsefusion_amd.scatter_points import (DynamicScatter, dynamic_point_to_voxel_forward,
                                             dynamic_scatter)
from sparsefusion_amd.voxelize import Voxelization
from test_oracle_scatter import make_case

pytestmark = pytest.mark.gpu


def check_vs_oracle(feats, coors, dev, oracle):
    f, c = torch.from_numpy(feats).to(dev), torch.from_numpy(coors).to(dev)
    for red in ('mean', 'max', 'sum'):
        ev, ec, em, ek = oracle.dynamic_scatter(feats, coors, red)
        vf, vc, m, k = [t.cpu().numpy() for t in dynamic_point_to_voxel_forward(f, c, red)]
        assert np.array_equal(vc, ec), red
        assert np.array_equal(m, em), red
        assert np.array_equal(k, ek), red
        if red == 'max':
            assert np.array_equal(vf.view(np.uint32), ev.view(np.uint32))
        else:
            assert np.allclose(vf, ev, rtol=1e-5, atol=2e-4 * max(1.0, float(np.abs(ev).max(initial=0.0)) / 50)), red


def test_reference_recipe(dev):
    # test_dynamic_scatter.py:12-51 with torch on the device as the checker
    torch.manual_seed(0)
    feats = torch.rand(size=(200000, 3), dtype=torch.float32, device=dev) * 100 - 50
    coors = torch.randint(low=-1, high=20, size=(200000, 3), dtype=torch.int32, device=dev)
    coors[coors.min(dim=-1).values < 0] = -1
    dsmean = DynamicScatter([0.32, 0.32, 6], [-74.88, -74.88, -2, 74.88, 74.88, 4], True)
    dsmax = DynamicScatter([0.32, 0.32, 6], [-74.88, -74.88, -2, 74.88, 74.88, 4], False)
    ref_coors = coors.unique(dim=0, sorted=True)
    ref_coors = ref_coors[ref_coors.min(dim=-1).values >= 0]
    lin = (coors[:, 0] * 400 + coors[:, 1] * 20 + coors[:, 2]).long()
    valid = coors.min(dim=-1).values >= 0
    ref_lin = (ref_coors[:, 0] * 400 + ref_coors[:, 1] * 20 + ref_coors[:, 2]).long()
    pos = torch.searchsorted(ref_lin, lin[valid])
    ssum = torch.zeros((ref_coors.shape[0], 3), dtype=torch.float64, device=dev)
    ssum.index_add_(0, pos, feats[valid].double())
    cnt = torch.bincount(pos, minlength=ref_coors.shape[0]).double()
    ref_mean = (ssum / cnt[:, None]).float()
    ref_max = torch.full((ref_coors.shape[0], 3), -float('inf'), device=dev)
    ref_max = ref_max.scatter_reduce(0, pos[:, None].expand(-1, 3), feats[valid], 'amax')
    feats_mean, coors_mean = dsmean(feats, coors)
    feats_max, coors_max = dsmax(feats, coors)
    assert (coors_mean == ref_coors).all()          # already in sorted order: no argsort needed
    assert torch.allclose(feats_mean, ref_mean, atol=1e-2, rtol=1e-5)
    assert (coors_max == ref_coors).all()
    assert torch.allclose(feats_max, ref_max, atol=1e-2, rtol=1e-5)
    assert repr(dsmean).startswith('DynamicScatter(voxel_size=[0.32, 0.32, 6]')


@pytest.mark.parametrize('seed,n,c,hi', [(0, 20000, 3, 20), (1, 5000, 4, 3), (2, 300, 5, 40),
                                         (3, 100000, 5, 64)])
def test_matches_oracle(dev, oracle, seed, n, c, hi):
    feats, coors = make_case(seed, n, c, hi)
    check_vs_oracle(feats, coors, dev, oracle)


def test_edges(dev, oracle):
    check_vs_oracle(np.ones((7, 2), np.float32), np.full((7, 3), -1, np.int32), dev, oracle)
    check_vs_oracle(np.zeros((0, 2), np.float32), np.zeros((0, 3), np.int32), dev, oracle)
    coors = np.array([[0, 0, 5], [0, 0, 5], [0, 1, 0], [-1, 3, 3]], np.int32)
    feats = np.array([[1, 2], [3, -4], [5, 6], [9, 9]], np.float32)
    check_vs_oracle(feats, coors, dev, oracle)
    with pytest.raises(RuntimeError):
        dynamic_scatter(torch.from_numpy(feats), torch.from_numpy(coors))          # host tensors
    with pytest.raises(RuntimeError):
        dynamic_scatter(torch.from_numpy(feats).to(dev), torch.from_numpy(coors).to(dev), 'min')


def test_full_frame_dynamic_voxelization(dev, oracle):
    # the dynamic-voxelization use: per-point coors from Voxelization(max_points=-1) -> mean per voxel
    pts = synth.nuscenes_frame(1)
    vl = configs.NUSC['pts_voxel_layer']
    layer = Voxelization(vl['voxel_size'], vl['point_cloud_range'], -1, -1)
    p = torch.from_numpy(pts).to(dev)
    coors = layer(p)
    ev, ec, em, ek = oracle.dynamic_scatter(pts, coors.cpu().numpy(), 'mean')
    vf, vc, m, k = [t.cpu().numpy() for t in dynamic_point_to_voxel_forward(p, coors, 'mean')]
    assert np.array_equal(vc, ec) and np.array_equal(m, em) and np.array_equal(k, ek)
    assert np.allclose(vf, ev, rtol=1e-5, atol=1e-3)
    # idempotence: scattering the voxel means with their own coordinates changes nothing
    vf2, vc2 = dynamic_scatter(torch.from_numpy(vf).to(dev), torch.from_numpy(vc).to(dev), 'mean')
    assert torch.equal(vc2.cpu(), torch.from_numpy(vc)) and torch.equal(vf2.cpu(), torch.from_numpy(vf))


def test_batched_coors(dev, oracle):
    fa, ca = make_case(5, 3000, 4, 12)
    fb, cb = make_case(6, 2000, 4, 12)
    ca, cb = ca[ca.min(-1) >= 0], cb[cb.min(-1) >= 0]
    fa, fb = fa[:ca.shape[0]], fb[:cb.shape[0]]
    coors = np.concatenate([np.pad(ca, ((0, 0), (1, 0)), constant_values=0),
                            np.pad(cb, ((0, 0), (1, 0)), constant_values=1)]).astype(np.int32)
    feats = np.concatenate([fa, fb])
    ds = DynamicScatter([1, 1, 1], [0, 0, 0, 12, 12, 12], True)
    vf, vc = ds(torch.from_numpy(feats).to(dev), torch.from_numpy(coors).to(dev))
    ea = oracle.dynamic_scatter(fa, ca, 'mean')
    eb = oracle.dynamic_scatter(fb, cb, 'mean')
    exp_c = np.concatenate([np.pad(ea[1], ((0, 0), (1, 0)), constant_values=0),
                            np.pad(eb[1], ((0, 0), (1, 0)), constant_values=1)])
    assert np.array_equal(vc.cpu().numpy(), exp_c)
    assert np.allclose(vf.cpu().numpy(), np.concatenate([ea[0], eb[0]]), rtol=1e-5, atol=2e-4)


def test_max_of_negative_zero_and_all_negative(dev):
    """A voxel whose only contribution to a channel is -0.0 must come out as -0.0, not the -inf
    the slot starts from (the reference's atomicMax on floats gives -0.0); all-negative channels
    give their largest value."""
    feats = np.asarray([[-0.0, -3.0, 1.0], [-2.5, -1.5, -0.0], [-7.0, -0.5, -4.0]], np.float32)
    coors = np.asarray([[0, 0, 1], [2, 1, 0], [2, 1, 0]], np.int32)
    f, c = torch.from_numpy(feats).to(dev), torch.from_numpy(coors).to(dev)
    vf, vc, m, k = [t.cpu().numpy() for t in dynamic_point_to_voxel_forward(f, c, 'max')]
    assert vc.tolist() == [[0, 0, 1], [2, 1, 0]]
    expect = np.asarray([[-0.0, -3.0, 1.0], [-2.5, -0.5, -0.0]], np.float32)
    assert np.array_equal(vf.view(np.uint32), expect.view(np.uint32))

"""torch.ops.sparsefusion.*: the reference's pybind entry points (voxel_layer, sparse_conv_ext) as
PyTorch custom ops over the C ABI.  Registration is checked on the CPU; results on the GPU against
the oracle (bit-exact for integer outputs, the fp32 bar for features)."""
import numpy as np
import pytest
import torch

from helpers import SMALL_RANGE, SMALL_SHAPE, SMALL_VOXEL, assert_close, small_points, to_dev


def test_ops_are_registered_for_the_device_key_only():
    from sparsefusion_amd import torch_ops
    for name in torch_ops.OPS:
        op = getattr(torch.ops.sparsefusion, name)
        assert op is not None
    with pytest.raises((NotImplementedError, RuntimeError)):      # no CPU kernel is registered
        torch.ops.sparsefusion.dynamic_voxelize(torch.zeros(4, 4), torch.zeros(4, 3, dtype=torch.int32),
                                                SMALL_VOXEL, SMALL_RANGE, 3)


@pytest.mark.gpu
def test_voxel_layer_ops(dev, oracle):
    from sparsefusion_amd import torch_ops  # noqa: F401
    pts = small_points(5, 4000)
    p = to_dev(pts, dev)
    mv, mp = 3000, 5
    voxels = torch.zeros((mv, mp, 5), device=dev)
    coors = torch.zeros((mv, 3), dtype=torch.int32, device=dev)
    num = torch.zeros((mv,), dtype=torch.int32, device=dev)
    n = torch.ops.sparsefusion.hard_voxelize(p, voxels, coors, num, SMALL_VOXEL, SMALL_RANGE, mp, mv, 3)
    ev, ec, en = oracle.hard_voxelize(pts, SMALL_VOXEL, SMALL_RANGE, mp, mv)
    assert n == ec.shape[0]
    assert np.array_equal(coors[:n].cpu().numpy(), ec) and np.array_equal(num[:n].cpu().numpy(), en)
    assert np.array_equal(voxels[:n].cpu().numpy(), ev)
    dc = torch.zeros((pts.shape[0], 3), dtype=torch.int32, device=dev)
    torch.ops.sparsefusion.dynamic_voxelize(p, dc, SMALL_VOXEL, SMALL_RANGE, 3)
    assert np.array_equal(dc.cpu().numpy(), oracle.dynamic_voxelize(pts, SMALL_VOXEL, SMALL_RANGE))
    vf, vc, p2v, cnt = torch.ops.sparsefusion.dynamic_point_to_voxel_forward(p, dc, 'max')
    evf, evc, ep2v, ecnt = oracle.dynamic_scatter(pts, dc.cpu().numpy(), 'max')
    assert np.array_equal(vc.cpu().numpy(), evc) and np.array_equal(vf.cpu().numpy(), evf)


@pytest.mark.gpu
@pytest.mark.parametrize('subm', [1, 0])
def test_sparse_conv_ext_ops(dev, oracle, subm):
    from sparsefusion_amd import torch_ops  # noqa: F401
    from test_gpu_rulebook import active_set
    rng = np.random.default_rng(4)
    idx = active_set(oracle, 9, 1800, 2)
    ks, st, pd = ([3, 3, 3], [1, 1, 1], [1, 1, 1]) if subm else ([3, 3, 3], [2, 2, 2], [1, 1, 1])
    e_out, e_pairs, e_num, e_shape = oracle.get_indice_pairs(idx, 2, SMALL_SHAPE, ks, st, pd, 1,
                                                             bool(subm))
    outids, pairs, num = torch.ops.sparsefusion.get_indice_pairs_3d(
        to_dev(idx, dev), 2, e_shape, SMALL_SHAPE, ks, st, pd, [1, 1, 1], [0, 0, 0], subm, 0)
    assert np.array_equal(outids.cpu().numpy(), e_out)
    assert np.array_equal(pairs.cpu().numpy(), e_pairs) and np.array_equal(num.cpu().numpy(), e_num)
    feats = rng.normal(0, 1, (idx.shape[0], 16)).astype(np.float32)
    w = rng.normal(0, 0.2, (3, 3, 3, 16, 32)).astype(np.float32)
    got = torch.ops.sparsefusion.indice_conv_fp32(to_dev(feats, dev), to_dev(w, dev), pairs, num,
                                                  outids.shape[0], 0, subm)
    expect = oracle.indice_conv(feats, w.reshape(27, 16, 32), e_pairs, e_num, e_out.shape[0],
                                bool(subm))
    assert_close(got.cpu().numpy(), expect, 1e-4)
    bev = torch.ops.sparsefusion.dense_bev(got, outids, 2, e_shape)
    assert np.array_equal(bev.cpu().numpy(), oracle.dense_bev(got.cpu().numpy(), e_out, 2, e_shape))

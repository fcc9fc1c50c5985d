"""sf_hard_voxelize_static -- the entry point that allocates nothing (static device scratch, for
callers under HIP graph capture or in loops that may not hipMalloc) -- against sf_hard_voxelize,
through the C ABI: bit-identical outputs on two calls in a row, the second with a smaller input
(stale scratch contents must not leak), then the limit check.

Collected last on purpose (the file name sorts after every other test file)."""
import numpy as np
import pytest
import torch

from sparsefusion_amd import _lib, configs, synth

pytestmark = pytest.mark.gpu


def run(L, dev, pts, vl, max_voxels, static, batch_idx=3):
    n, f = pts.shape
    mp = vl['max_num_points']
    voxels = torch.full((max_voxels, mp, f), -7.0, dtype=torch.float32, device=dev)
    coors = torch.full((max_voxels, 4), -7, dtype=torch.int32, device=dev)
    npv = torch.full((max_voxels,), -7, dtype=torch.int32, device=dev)
    mean = torch.full((max_voxels, 8), -7.0, dtype=torch.float32, device=dev)
    count = torch.zeros(2, dtype=torch.int32, device=dev)       # [row count, overflow]
    p = torch.from_numpy(pts).to(dev)
    head = (_lib.ptr(p), n, f, _lib.f3(vl['voxel_size']), _lib.f6(vl['point_cloud_range']), mp,
            max_voxels, voxels.data_ptr(), coors.data_ptr(), 4, batch_idx, npv.data_ptr(),
            mean.data_ptr(), 5, 8, count.data_ptr(), max_voxels, count.data_ptr() + 4)
    st = _lib.stream(dev)
    if static:
        rc = L.sf_hard_voxelize_static(*head, st)
    else:
        wsb = L.sf_hard_voxelize_workspace(n, mp)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        rc = L.sf_hard_voxelize(*head, None, ws.data_ptr(), wsb, st)
    _lib.check(rc)
    torch.cuda.synchronize()
    m, ovf = count.tolist()
    return m, ovf, voxels.cpu().numpy(), coors.cpu().numpy(), npv.cpu().numpy(), mean.cpu().numpy()


def test_static_entry_point_matches_workspace_entry_point(dev):
    L = _lib.lib()
    vl = configs.NUSC['pts_voxel_layer']
    big = synth.nuscenes_frame(2)                  # ~320 k points, near the limit
    small = np.ascontiguousarray(synth.nuscenes_frame(3)[:41234])
    assert L.sf_hard_voxelize_workspace(big.shape[0], 10) <= (32 << 20)
    for pts, mv in ((big, 160000), (small, 9000)):  # second call: smaller input, cap triggers
        ref = run(L, dev, pts, vl, mv, static=False)
        got = run(L, dev, pts, vl, mv, static=True)
        assert got[0] == ref[0] > 0 and got[1] == ref[1] == 0
        for a, b in zip(got[2:], ref[2:]):          # whole buffers, untouched rows included
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert ref[0] == 9000


def test_static_entry_point_under_graph_capture_allocates_nothing(dev):
    """The two claims of the header: the call can be CAPTURED in a HIP graph (after one eager call on
    the device, which resolves the scratch symbol) and replayed on new buffer contents with the
    results of sf_hard_voxelize, and it takes no device memory (hipMemGetInfo unchanged)."""
    L = _lib.lib()
    vl = configs.NUSC['pts_voxel_layer']
    mp, mv, f = vl['max_num_points'], 20000, 5
    frames = [np.ascontiguousarray(synth.nuscenes_frame(s)[:60000]) for s in (4, 5)]
    n = frames[0].shape[0]
    p = torch.from_numpy(frames[0]).to(dev)
    voxels = torch.zeros((mv, mp, f), dtype=torch.float32, device=dev)
    coors = torch.zeros((mv, 4), dtype=torch.int32, device=dev)
    npv = torch.zeros((mv,), dtype=torch.int32, device=dev)
    mean = torch.zeros((mv, 8), dtype=torch.float32, device=dev)
    count = torch.zeros(2, dtype=torch.int32, device=dev)
    args = (_lib.ptr(p), n, f, _lib.f3(vl['voxel_size']), _lib.f6(vl['point_cloud_range']), mp, mv,
            voxels.data_ptr(), coors.data_ptr(), 4, 0, npv.data_ptr(), mean.data_ptr(), 5, 8,
            count.data_ptr(), mv, count.data_ptr() + 4)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        _lib.check(L.sf_hard_voxelize_static(*args, _lib.stream(dev)))      # eager warm-up call
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    with torch.cuda.stream(side):
        for _ in range(3):
            count.zero_()
            _lib.check(L.sf_hard_voxelize_static(*args, _lib.stream(dev)))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(dev)[0] == free0          # nothing allocated, nothing freed
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        count.zero_()
        _lib.check(L.sf_hard_voxelize_static(*args, _lib.stream(dev)))
    for pts in (frames[1], frames[0]):                        # replays on changed contents
        p.copy_(torch.from_numpy(pts))
        for t in (voxels, coors, npv, mean):
            t.fill_(-7)
        g.replay()
        torch.cuda.synchronize()
        got = (count.tolist(), voxels.cpu().numpy(), coors.cpu().numpy(), npv.cpu().numpy(),
               mean.cpu().numpy())
        m, ovf, *ref = run(L, dev, pts, vl, mv, static=False, batch_idx=0)
        assert got[0] == [m, ovf] and m > 0
        for a, b in zip(got[1:], ref):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_static_entry_point_rejects_oversized_input(dev):
    L = _lib.lib()
    vl = configs.WAYMO['pts_voxel_layer']
    n = 1000000
    assert L.sf_hard_voxelize_workspace(n, 5) > (32 << 20)
    p = torch.zeros((n, 5), dtype=torch.float32, device=dev)
    out = torch.zeros(16, dtype=torch.int32, device=dev)
    rc = L.sf_hard_voxelize_static(_lib.ptr(p), n, 5, _lib.f3(vl['voxel_size']),
                                   _lib.f6(vl['point_cloud_range']), 5, 10, None, out.data_ptr(), 3,
                                   0, None, None, 0, 0, out.data_ptr() + 48, 1, None,
                                   _lib.stream(dev))
    assert rc == -2 and b'static buffer holds' in L.sf_last_error()

"""sf_row_order_by_mask + sf_sparse_conv_f32_ordered (csrc/spconv_f32_ordered.hip): the mask-ordered,
register-accumulator decomposition of the fp32 sparse convolution, against the CPU oracle
(indiceConv restatement) and against sf_sparse_conv_f32 on the same rulebook.  fp32 bar of the
other conv tests: |a-b| <= 1e-4|b| + 1e-5 max|b|."""
import numpy as np
import pytest
import torch

from helpers import SMALL_RANGE, SMALL_SHAPE, SMALL_VOXEL, assert_close, small_points, to_dev
from sparsefusion_amd import _lib, spconv

pytestmark = pytest.mark.gpu


def _coords(oracle, seed, n, batch=1):
    cs = [oracle.hard_voxelize(small_points(seed + b, n, 4), SMALL_VOXEL, SMALL_RANGE, 5, 20000)[1]
          for b in range(batch)]
    return oracle.batch_coors(cs)


def _row_order(rb, dev):
    L = _lib.lib()
    n = rb.out_coords.n
    order = torch.empty(L.sf_row_order_size(max(n, 1)), dtype=torch.int32, device=dev)
    wsb = L.sf_row_order_workspace(max(n, 1))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.check(L.sf_row_order_by_mask(rb.nbr.data_ptr(), rb.nbr.shape[1], rb.K, rb.k_first, None, n,
                                      order.data_ptr(), ws.data_ptr(), wsb, _lib.stream(dev)))
    return order


@pytest.mark.parametrize('subm,ks,st,pd', [(True, 3, 1, 1), (False, 3, 2, 1), (False, (3, 1, 1), (2, 1, 1), 0)])
def test_row_order_is_a_mask_sorted_permutation(dev, oracle, subm, ks, st, pd):
    coors = _coords(oracle, 40, 6000, 2)
    cs = spconv.CoordSet(to_dev(coors, dev), SMALL_SHAPE, 2)
    rb = spconv.build_rulebook(cs, ks, st, pd, 1, subm)
    order = _row_order(rb, dev).cpu().numpy()
    n = rb.out_coords.n
    assert np.array_equal(np.sort(order[:n]), np.arange(n))
    nbr = rb.nbr[:, :n].cpu().numpy()
    mask = ((nbr >= 0).astype(np.int64) << np.arange(rb.K)[:, None]).sum(0)
    m = mask[order[:n]]
    assert np.all(np.diff(m) >= 0)                                   # sorted by mask
    same = np.diff(m) == 0
    assert np.all(np.diff(order[:n])[same] > 0)                       # stable inside a mask
    # the tile schedule behind the order: every 64-row tile once, by descending work
    tiles = (n + 63) // 64
    sched = order[n:n + tiles]
    assert np.array_equal(np.sort(sched), np.arange(tiles))
    pad = np.zeros(tiles * 64, np.int64)
    pad[:n] = m
    blocks = np.bitwise_or.reduce(pad.reshape(-1, 16), axis=1)
    work = np.array([bin(int(b)).count('1') for b in blocks]).reshape(tiles, 4).sum(1)
    nph = np.array([bin(int(np.bitwise_or.reduce(pad[t * 64:(t + 1) * 64]))).count('1') for t in range(tiles)])
    cost = work + 3 * nph                      # a phase costs about three block products of fixed work
    assert np.all(np.diff(cost[sched]) <= 0)
    # tile headers (the tile's phases ready made, in execution order: the centre offset first when it
    # is active, then the active offsets ascending, padded to three phases) and the permuted table
    hdr = order[n + tiles:n + tiles + 32 * tiles].reshape(tiles, 32)
    tab = order[n + 33 * tiles:n + 33 * tiles + 27 * 64 * tiles].reshape(27, tiles * 64)
    want = np.full((27, tiles * 64), -1, np.int32)
    want[:rb.K, :n] = nbr[:, order[:n]]
    assert np.array_equal(tab[:rb.K], want[:rb.K])
    kf = rb.k_first
    for t in range(tiles):
        bm = [(want[k, t * 64:(t + 1) * 64].reshape(4, 16) >= 0).any(1) for k in range(rb.K)]
        act = [(k, int(sum(int(v) << i for i, v in enumerate(bm[k])))) for k in range(rb.K) if bm[k].any()]
        if kf >= 0 and any(k == kf for k, _ in act):
            act = [a for a in act if a[0] == kf] + [a for a in act if a[0] != kf]
        npad = max(len(act), 3)
        words = []
        for pos in range(npad):
            flags = (2 if pos == 0 else 0) | (4 if pos == npad - 1 else 0) | (8 if pos == 1 else 0) | 16
            w = flags << 9
            if pos < len(act):
                w |= act[pos][0] | (act[pos][1] << 5) | (1 << 9)
            words.append(w)
        assert hdr[t, 0] == npad and list(hdr[t, 1:1 + npad]) == words, t
        assert hdr[t, 28] == cost[t]


@pytest.mark.parametrize('cin,cout,subm,ks,st,pd', [
    (16, 16, True, 3, 1, 1), (16, 32, False, 3, 2, 1), (32, 32, True, 3, 1, 1),
    (32, 64, False, 3, 2, 1), (64, 64, True, 3, 1, 1), (64, 128, False, 3, 2, (0, 1, 1)),
    (128, 128, True, 3, 1, 1), (128, 128, False, (3, 1, 1), (2, 1, 1), 0), (128, 64, True, 3, 1, 1)])
def test_ordered_conv_matches_oracle(dev, oracle, cin, cout, subm, ks, st, pd):
    rng = np.random.default_rng(cin + cout)
    coors = _coords(oracle, 50 + cin, 5000)
    n_in = coors.shape[0]
    feats = rng.normal(0, 1, (n_in, cin)).astype(np.float32)
    tri = lambda v: list(v) if isinstance(v, (tuple, list)) else [v] * 3
    K = int(np.prod(tri(ks)))
    w = (rng.uniform(-1, 1, (K, cin, cout)) * np.sqrt(3.0 / (min(K, 9) * cin))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.1, cout).astype(np.float32)
    outids, pairs, num, _ = oracle.get_indice_pairs(coors, 1, SMALL_SHAPE, ks, st, pd, 1, subm)
    n_out = outids.shape[0]
    res = rng.normal(0, 1, (n_out, cout)).astype(np.float32) if subm else None
    y = oracle.indice_conv(feats, w, pairs, num, n_out, subm)
    expect = y * scale + shift + (res if res is not None else 0)
    expect = np.maximum(expect, 0).astype(np.float32)
    cs = spconv.CoordSet(to_dev(coors, dev), SMALL_SHAPE, 1)
    rb = spconv.build_rulebook(cs, ks, st, pd, 1, subm)
    assert np.array_equal(rb.out_coords.indices[:n_out].cpu().numpy(), outids)
    L = _lib.lib()
    d_feats, d_w = to_dev(feats, dev), spconv.pack_filters(to_dev(w, dev))
    d_scale, d_shift = to_dev(scale, dev), to_dev(shift, dev)
    d_res = to_dev(res, dev) if res is not None else None
    order = _row_order(rb, dev)
    out = torch.full((n_out, cout), np.nan, dtype=torch.float32, device=dev)
    _lib.check(L.sf_sparse_conv_f32_ordered(
        d_feats.data_ptr(), n_in, cin, rb.nbr.data_ptr(), rb.nbr.shape[1], rb.K, rb.k_first,
        order.data_ptr(), d_w.data_ptr(), cout, d_scale.data_ptr(),
        d_shift.data_ptr(), _lib.ptr(d_res), 1, out.data_ptr(), None, n_out, _lib.stream(dev)))
    torch.cuda.synchronize()
    assert_close(out.cpu().numpy(), expect)
    ref = torch.empty_like(out)
    _lib.check(L.sf_sparse_conv_f32(d_feats.data_ptr(), n_in, cin, rb.nbr.data_ptr(), rb.nbr.shape[1], rb.K,
                                    rb.k_first, d_w.data_ptr(), cout, d_scale.data_ptr(),
                                    d_shift.data_ptr(), _lib.ptr(d_res), 1, ref.data_ptr(), None,
                                    n_out, _lib.stream(dev)))
    torch.cuda.synchronize()
    assert_close(out.cpu().numpy(), ref.cpu().numpy())


@pytest.mark.parametrize('cin,cout,cap,batch', [(64, 64, 1, 1), (128, 128, 1, 1), (64, 64, 3, 2),
                                                (32, 32, 1, 1)])
def test_many_tiles_per_workgroup(dev, oracle, cin, cout, cap, batch):
    """The persistent workgroups walk their tiles in runs of 128 (kRun): small frames give every
    workgroup one tile, a full-size frame two or three, so the second-run path is reached only with
    the grid capped.  sf_debug_ordered_grid_cap (diagnostic hook) limits the grid to `cap`
    workgroups: > 128 tiles each here; the result must not change."""
    import ctypes
    rng = np.random.default_rng(900 + cin)
    coors = _coords(oracle, 61, 20000, batch)
    n = coors.shape[0]
    assert (n + 63) // 64 > 128 * cap + 2, n
    feats = rng.normal(0, 1, (n, cin)).astype(np.float32)
    w = (rng.uniform(-1, 1, (27, cin, cout)) * np.sqrt(3.0 / (9 * cin))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.1, cout).astype(np.float32)
    outids, pairs, num, _ = oracle.get_indice_pairs(coors, batch, SMALL_SHAPE, 3, 1, 1, 1, True)
    expect = np.maximum(oracle.indice_conv(feats, w, pairs, num, n, True) * scale + shift, 0)
    cs = spconv.CoordSet(to_dev(coors, dev), SMALL_SHAPE, batch)
    rb = spconv.build_rulebook(cs, 3, 1, 1, 1, True)
    L = _lib.lib()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    raw.sf_debug_ordered_grid_cap.argtypes = [ctypes.c_int]
    d_feats, d_w = to_dev(feats, dev), spconv.pack_filters(to_dev(w, dev))
    d_scale, d_shift = to_dev(scale, dev), to_dev(shift, dev)
    order = _row_order(rb, dev)
    outs = []
    for c in (0, cap):
        out = torch.full((n, cout), np.nan, dtype=torch.float32, device=dev)
        raw.sf_debug_ordered_grid_cap(c)
        try:
            _lib.check(L.sf_sparse_conv_f32_ordered(
                d_feats.data_ptr(), n, cin, rb.nbr.data_ptr(), rb.nbr.shape[1], rb.K, rb.k_first,
                order.data_ptr(), d_w.data_ptr(), cout, d_scale.data_ptr(), d_shift.data_ptr(), None,
                1, out.data_ptr(), None, n, _lib.stream(dev)))
            torch.cuda.synchronize()
        finally:
            raw.sf_debug_ordered_grid_cap(0)
        outs.append(out.cpu().numpy())
    assert_close(outs[1], expect.astype(np.float32))
    assert np.array_equal(outs[0], outs[1])          # same sums in the same order, whatever the grid


def test_leftover_tiles_dealt_as_halves(dev, oracle):
    """The 128 -> 128 kernel (one workgroup per CU) deals the tiles left over after the whole rounds
    of its schedule as HALF tiles (16-row blocks 0-1 / 2-3), as many as it takes to give every
    workgroup an entry.  Grid caps from 3 to 64 workgroups over one problem hit every pattern --
    nothing left over, all leftovers split (rem <= G / 2), whole and split leftovers mixed -- with
    residual and ReLU; every grid must give the SAME bits (the summation order per output element does
    not depend on the schedule) and the oracle's values."""
    import ctypes
    rng = np.random.default_rng(4128)
    cin = cout = 128
    coors = _coords(oracle, 77, 4000)
    n = coors.shape[0]
    tiles = (n + 63) // 64
    feats = rng.normal(0, 1, (n, cin)).astype(np.float32)
    res = rng.normal(0, 1, (n, cout)).astype(np.float32)
    w = (rng.uniform(-1, 1, (27, cin, cout)) * np.sqrt(3.0 / (9 * cin))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.1, cout).astype(np.float32)
    outids, pairs, num, _ = oracle.get_indice_pairs(coors, 1, SMALL_SHAPE, 3, 1, 1, 1, True)
    expect = np.maximum(oracle.indice_conv(feats, w, pairs, num, n, True) * scale + shift + res, 0)
    cs = spconv.CoordSet(to_dev(coors, dev), SMALL_SHAPE, 1)
    rb = spconv.build_rulebook(cs, 3, 1, 1, 1, True)
    L = _lib.lib()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    raw.sf_debug_ordered_grid_cap.argtypes = [ctypes.c_int]
    d_feats, d_w, d_res = to_dev(feats, dev), spconv.pack_filters(to_dev(w, dev)), to_dev(res, dev)
    d_scale, d_shift = to_dev(scale, dev), to_dev(shift, dev)
    order = _row_order(rb, dev)
    caps = [0, 3, 4, 5, 7, 8, 16, 33, 64] + [c for c in (tiles, tiles - 1, (tiles + 1) // 2) if c > 2]
    patterns, first = set(), None
    for c in caps:
        g = min(c, 256) if c else min(tiles, 256)
        rem = tiles % g
        patterns.add('none' if rem == 0 else ('all split' if 2 * rem <= g else 'mixed'))
        out = torch.full((n, cout), np.nan, dtype=torch.float32, device=dev)
        raw.sf_debug_ordered_grid_cap(c)
        try:
            _lib.check(L.sf_sparse_conv_f32_ordered(
                d_feats.data_ptr(), n, cin, rb.nbr.data_ptr(), rb.nbr.shape[1], rb.K, rb.k_first,
                order.data_ptr(), d_w.data_ptr(), cout, d_scale.data_ptr(), d_shift.data_ptr(),
                d_res.data_ptr(), 1, out.data_ptr(), None, n, _lib.stream(dev)))
            torch.cuda.synchronize()
        finally:
            raw.sf_debug_ordered_grid_cap(0)
        assert L.sf_last_conv_kernel().decode().startswith('conv_f32_dma_kernel<128, 128')
        got = out.cpu().numpy()
        if first is None:
            first = got
            assert_close(got, expect.astype(np.float32))
        assert np.array_equal(got, first), c
    assert patterns == {'none', 'all split', 'mixed'}, patterns


def test_pipeline_with_the_ordered_kernel(dev, oracle, monkeypatch):
    """LidarBackbone with every eligible layer on the ordered kernel (spconv.ORDERED_SHAPES = 'all'), device
    row counts, captured in a HIP graph: BEV map against the oracle."""
    from sparsefusion_amd.pipeline import LidarBackbone
    from test_gpu_conv import make_encoder
    from test_gpu_pipeline import VL, oracle_bev
    from sparsefusion_amd import spconv
    monkeypatch.setattr(spconv, 'ORDERED_SHAPES', 'all')
    enc, sd, cfg = make_encoder(dev, 5, 21)
    frames = [small_points(700 + b, 3000 + 800 * b) for b in range(2)]
    pts = [to_dev(p, dev) for p in frames]
    bb = LidarBackbone(VL, enc, dev, max_points=8000, batch_size=2)
    assert any(l.ordered for l in bb.layers) and not bb.layers[0].ordered
    with torch.no_grad():
        g = bb.capture(pts)
        g.replay()
        torch.cuda.synchronize()
        bev = bb.bev.clone()
    assert bb.results()['overflow'] == 0
    expect, _, _ = oracle_bev(oracle, sd, cfg, frames, 10, 6000)
    assert_close(bev.cpu().numpy(), expect, 1e-4)


def test_bad_arguments(dev):
    L = _lib.lib()
    x = torch.zeros(64, 16, device=dev)
    nbr = torch.full((27, 64), -1, dtype=torch.int32, device=dev)
    assert L.sf_sparse_conv_f32_ordered(x.data_ptr(), 64, 8, nbr.data_ptr(), 64, 27, 13, nbr.data_ptr(),
                                        x.data_ptr(), 16, None, None, None, 0, x.data_ptr(), None, 64,
                                        None) < 0
    assert b'cin_padded' in L.sf_last_error()
    assert L.sf_sparse_conv_f32_ordered(x.data_ptr(), 64, 16, nbr.data_ptr(), 64, 27, 13, None,
                                        x.data_ptr(), 16, None, None, None, 0, x.data_ptr(), None, 64,
                                        None) < 0       # the plan of sf_row_order_by_mask is required
    assert L.sf_row_order_by_mask(nbr.data_ptr(), 64, 27, 13, None, 64, x.data_ptr(), x.data_ptr(), 16,
                                  None) < 0
    assert b'workspace' in L.sf_last_error()

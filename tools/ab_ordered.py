#!/usr/bin/env python3
"""A/B of the two fp32 sparse-conv decompositions on every layer of a real frame:
sf_sparse_conv_f32 (pair compaction / direct) against sf_sparse_conv_f32_ordered (mask-ordered rows,
register accumulators, empty blocks skipped), same inputs, same rulebooks.

    python tools/ab_ordered.py [--frame 0] [--batch 1] [--reps 10] [--natural]

Prints per layer: rows, pairs, time of each, max |a-b| / max|a| between them."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sparsefusion_amd import _lib, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frame', type=int, default=0)
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--natural', action='store_true', help='no row order (natural order)')
    ap.add_argument('--wide-order', action='store_true',
                    help='B side = the compacting wide kernel of sf_sparse_conv_f32 with the mask row order')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    pts = [synth.nuscenes_frame(args.frame + b) for b in range(args.batch)]
    cfg, ecfg, sd, enc, bb = bench.build_backbone('nusc', dev, args.batch, max(p.shape[0] for p in pts),
                                                  overlap=False)
    f = [torch.from_numpy(p).to(dev) for p in pts]
    L = _lib.lib()
    st = _lib.stream(dev)
    real = L.sf_sparse_conv_f32
    orders = {}
    rows = []
    state = {'i': 0}

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps * 1e3

    class Shim:
        def __getattr__(self, n):
            return getattr(L, n)

    def wrapped(*a):
        (feat, cinp, nbr, stride, K, kf, wpk, cout, scale, shift, res, relu, out, n_dev, n_bound,
         stream) = a
        i = state['i']
        state['i'] += 1
        lay = bb.layers[i]
        rc = real(*a)
        if cinp < 16:
            rows.append((i, lay.cin, cout, K, None))
            return rc
        torch.cuda.synchronize()
        n_in_bound = bb.sets[lay.rb['src']]['bound']
        n_out = int(bb.counts[lay.rb['dst']].item())
        key = nbr
        if key not in orders:
            order = torch.empty(n_bound, dtype=torch.int32, device=dev)
            wsb = L.sf_row_order_workspace(n_bound)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            t_order = timeit(lambda: _lib.check(L.sf_row_order_by_mask(
                nbr, stride, K, kf, n_dev, n_bound, order.data_ptr(), ws.data_ptr(), wsb, st)))
            orders[key] = (order, t_order)
        order, t_order = orders[key]
        ref = torch.empty((n_bound, cout), dtype=torch.float32, device=dev)
        alt = torch.empty((n_bound, cout), dtype=torch.float32, device=dev)
        t_a = timeit(lambda: _lib.check(real(feat, cinp, nbr, stride, K, kf, wpk, cout, scale, shift,
                                             res, relu, ref.data_ptr(), n_dev, n_bound, stream)))
        optr = None if args.natural else order.data_ptr()
        if args.wide_order:
            import ctypes
            raw = ctypes.CDLL(_lib.LIB_PATH)
            raw.sf_debug_conv_row_order.argtypes = [ctypes.c_void_p]

            def run_b():
                raw.sf_debug_conv_row_order(order.data_ptr())
                rc_ = real(feat, cinp, nbr, stride, K, kf, wpk, cout, scale, shift, res, relu,
                           alt.data_ptr(), n_dev, n_bound, stream)
                raw.sf_debug_conv_row_order(None)
                _lib.check(rc_)
            t_b = timeit(run_b)
        else:
          t_b = timeit(lambda: _lib.check(L.sf_sparse_conv_f32_ordered(
            feat, n_in_bound, cinp, nbr, stride, K, kf, optr, wpk, cout, scale, shift, res, relu,
            alt.data_ptr(), n_dev, n_bound, stream)))
        a_, b_ = ref[:n_out], alt[:n_out]
        err = float((a_ - b_).abs().max() / a_.abs().max().clamp_min(1e-30))
        rows.append((i, lay.cin, cout, K, (n_out, t_a, t_b, err, t_order)))
        return rc

    shim = Shim()
    shim.__dict__['sf_sparse_conv_f32'] = wrapped
    with torch.no_grad():
        bb.enqueue(f)
        torch.cuda.synchronize()
        bb.L = shim
        bb.enqueue(f)
        torch.cuda.synchronize()
    ta = tb = 0.0
    for i, cin, cout, K, r in rows:
        if r is None:
            print(f'{i:2d} {cin:3d}->{cout:3d} K{K:2d}  (first layer: not covered by the ordered kernel)')
            continue
        n_out, t_a, t_b, err, t_o = r
        ta += t_a
        tb += t_b
        print(f'{i:2d} {cin:3d}->{cout:3d} K{K:2d} rows {n_out:7d}  current {t_a:7.1f} us  ordered {t_b:7.1f} us '
              f'({t_a / t_b:4.2f}x)  rel diff {err:.1e}  [order build {t_o:.0f} us]')
    print(f'sum: current {ta:.0f} us, ordered {tb:.0f} us; row orders {sum(o[1] for o in orders.values()):.0f} us')


if __name__ == '__main__':
    main()

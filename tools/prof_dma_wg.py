#!/usr/bin/env python3
"""How level do the persistent workgroups of the LDS-DMA conv kernels finish?  Every launch of one
eager frame stamps the 100 MHz wall clock at the start and end of each workgroup
(sf_debug_dma_wg_times); printed per launch: kernel span, mean / percentiles of the workgroups'
end times relative to the first start, and the share of workgroup-time spent waiting for the
slowest one (1 - mean busy / span).

    python tools/prof_dma_wg.py [--frame 0]
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sparsefusion_amd import _lib, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frame', type=int, default=0)
    ap.add_argument('--grid-cap', type=int, default=0, help='limit the persistent grid (sf_debug_ordered_grid_cap)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    pts = synth.nuscenes_frame(args.frame)
    cfg, ecfg, sd, enc, bb = bench.build_backbone('nusc', dev, 1, pts.shape[0])
    f = [torch.from_numpy(pts).to(dev)]
    L = _lib.lib()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    raw.sf_debug_dma_wg_times.argtypes = [ctypes.c_void_p]
    if args.grid_cap:
        raw.sf_debug_ordered_grid_cap(args.grid_cap)
    buf = torch.zeros(2 * 4096, dtype=torch.int64, device=dev)
    with torch.no_grad():
        for _ in range(3):
            bb.enqueue(f)
        torch.cuda.synchronize()
        real = L.sf_sparse_conv_f32_ordered

        class Shim:
            def __getattr__(self, n):
                return getattr(L, n)
        res = []

        def wrapped(*a):
            buf.zero_()
            torch.cuda.synchronize()
            raw.sf_debug_dma_wg_times(buf.data_ptr())
            rc = real(*a)
            torch.cuda.synchronize()
            raw.sf_debug_dma_wg_times(None)
            res.append((a[2], a[9], L.sf_last_conv_kernel().decode(), buf.cpu().numpy().copy(), a[7], a[16]))
            return rc
        sh = Shim()
        sh.__dict__['sf_sparse_conv_f32_ordered'] = wrapped
        bb.L = sh
        bb.enqueue(f)
        torch.cuda.synchronize()
    plans = {}
    for lay in bb.layers:
        if getattr(lay, 'ordered', False):
            plans[lay.rb['order'].data_ptr()] = (lay.rb['order'].cpu().numpy(), int(bb.counts[lay.rb['dst']].item()))
    for cin, cout, kern, v, plan_ptr, n_bound in res:
        g = 256 if '128, 128' in kern else 768       # resident workgroups of the two instantiations
        if args.grid_cap:
            g = min(g, args.grid_cap)
        st, en = v[:g].astype(np.float64), v[g:2 * g].astype(np.float64)
        t0 = st.min()
        span = (en.max() - t0) / 100.0
        busy = (en - st) / 100.0
        endr = (en - t0) / 100.0
        print(f'{kern}: grid {g} span {span:.1f} us | start spread {(st.max() - t0) / 100:.1f} us | '
              f'end p10 {np.percentile(endr, 10):.1f} p50 {np.percentile(endr, 50):.1f} p90 {np.percentile(endr, 90):.1f} '
              f'max {endr.max():.1f} | mean busy {busy.mean():.1f} us = {busy.mean() / span:.1%} of the span')
        # the plan's view of the same launch: products and phases of every workgroup, and the least-squares
        # fit busy = a * products + b * phases + c (c is an extrapolation far outside the data: all
        # workgroups hold about the same work)
        plan, n = plans[plan_ptr]
        tiles_b, tiles = (n_bound + 63) // 64, (n + 63) // 64
        print(f'    rows {n}, tiles {tiles}: {tiles // g} whole rounds on {g} workgroups + {tiles % g} tiles left over')
        sched = plan[n_bound:n_bound + tiles_b]
        hdr = plan[n_bound + tiles_b:n_bound + tiles_b + 32 * tiles_b].reshape(tiles_b, 32)
        nact_t = np.array([bin(int(m) & 0x7FFFFFF).count('1') for m in hdr[:, 29]])
        prod_t = hdr[:, 28] - 3 * nact_t
        ph_t = hdr[:, 0]                 # phases incl. padding
        wp, wph = np.zeros(g), np.zeros(g)
        for i in range((tiles + g - 1) // g):
            for b in range(g):
                j = i * g + ((g - 1 - b) if (i & 1) else b)
                if j < tiles:
                    t = sched[j]
                    wp[b] += prod_t[t]
                    wph[b] += ph_t[t]
        A = np.stack([wp, wph, np.ones(g)], 1)
        coef, *_ = np.linalg.lstsq(A, busy, rcond=None)
        pred = A @ coef
        print(f'    plan: products per workgroup mean {wp.mean():.1f} max {wp.max():.0f}, phases mean {wph.mean():.1f} max {wph.max():.0f} | '
              f'fit busy = {coef[0]:.3f} us x products + {coef[1]:.3f} us x phases + {coef[2]:.1f} us (rms error {np.sqrt(((pred - busy) ** 2).mean()):.1f} us)')


if __name__ == '__main__':
    main()

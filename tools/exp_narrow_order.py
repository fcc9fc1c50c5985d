#!/usr/bin/env python3
"""Experiment (not product code): how much faster is the narrow 32->32 / 16->16 kernel when the rows
of its stage come in (a) natural order, (b) global mask order, (c) mask order inside chunks of
2048 rows?  The coordinate set of the stage is permuted on the host and the rulebook rebuilt, so the
kernel is unchanged; the time of sf_sparse_conv_f32 alone is measured with HIP events.

    python tools/exp_narrow_order.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sparsefusion_amd import _lib, spconv, synth  # noqa: E402


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device('cuda:0')
    pts = synth.nuscenes_frame(0)
    cfg, ecfg, sd, enc, bb = bench.build_backbone('nusc', dev, 1, pts.shape[0])
    with torch.no_grad():
        bb.enqueue([torch.from_numpy(pts).to(dev)])
    torch.cuda.synchronize()
    counts = bb.counts.tolist()
    for si, (cin, cout) in ((0, (16, 16)), (1, (32, 32))):
        n = counts[si]
        shape = bb.sets[si]['shape']
        coords = bb.sets[si]['indices'][:n].clone()
        # masks from the pipeline's own SubM table of this set
        rb0 = [r for r in bb.rulebooks if r['subm'] and r['dst'] == si][0]
        nbr = rb0['nbr'][:, :n].cpu().numpy()
        mask = ((nbr >= 0).astype(np.int64) << np.arange(27)[:, None]).sum(0)
        orders = {'natural': np.arange(n), 'global mask order': np.argsort(mask, kind='stable')}
        for ch in (2048, 8192):
            orders[f'mask order in chunks of {ch}'] = np.concatenate(
                [s + np.argsort(mask[s:s + ch], kind='stable') for s in range(0, n, ch)])
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        packed = spconv.pack_filters(w) if hasattr(spconv, 'pack_filters') else None
        for name, order in orders.items():
            c = coords[torch.from_numpy(order).to(dev)].contiguous()
            cs = spconv.CoordSet(c, shape, 1)
            rb = spconv.build_rulebook(cs, 3, 1, 1, 1, True)
            m2 = ((rb.nbr[:, :n].cpu().numpy() >= 0).astype(np.int64) << np.arange(27)[:, None]).sum(0)
            blocks = np.bitwise_or.reduce(np.pad(m2, (0, (-n) % 16)).reshape(-1, 16), axis=1)
            issued = sum(int(((blocks >> k) & 1).sum()) for k in range(27)) * 16
            pairs = sum(int(((m2 >> k) & 1).sum()) for k in range(27))
            x = torch.randn(n, cin, device=dev)
            conv = spconv.SubMConv3d(cin, cout, 3, bias=False).to(dev)
            t = spconv.SparseConvTensor(x, c, shape, 1)
            t.coord_set = cs if hasattr(t, 'coord_set') else None
            with torch.no_grad():
                us = timed(lambda: conv(t))
            print(f'stage {si + 1} {cin}->{cout} rows {n} | {name:28s} | MFMA rows issued / pairs {issued / pairs:.2f} | '
                  f'{us:7.1f} us per SubMConv3d forward ({_lib.lib().sf_last_conv_kernel().decode()})', flush=True)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Per-layer timing of SECOND + SECONDFPN (sf_conv2d_f32) on the target config's shapes.

    python tools/bench_second.py [--batch B] [--iters N] [--json out.json]

For every distinct layer shape: mean launch duration (HIP events on the launch stream), algorithmic
flops 2*B*Ho*Wo*Cout*Cin*k*k and TFLOP/s against the fp32 MFMA peak (157.3).  Also the whole
backbone + neck per BEV map, eager and from a HIP graph."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sparsefusion_amd import _lib, configs, synth  # noqa: E402
from sparsefusion_amd import second as S  # noqa: E402

PEAK = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--size', type=int, default=180)
    ap.add_argument('--json', default='')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    bcfg, ncfg = dict(configs.NUSC['pts_backbone']), dict(configs.NUSC['pts_neck'])
    bcfg.pop('type'), ncfg.pop('type')
    second, neck = S.SECOND(**bcfg), S.SECONDFPN(**ncfg)
    second.load_state_dict({k: torch.from_numpy(np.asarray(v))
                            for k, v in synth.second_state_dict(1, **bcfg).items()})
    neck.load_state_dict({k: torch.from_numpy(np.asarray(v))
                          for k, v in synth.second_fpn_state_dict(2, **ncfg).items()})
    second, neck = second.to(dev).eval(), neck.to(dev).eval()
    rng = np.random.default_rng(0)
    x = torch.from_numpy((rng.normal(0, 1, (args.batch, 256, args.size, args.size)) *
                          (rng.random((args.batch, 1, args.size, args.size)) < 0.4)
                          ).astype(np.float32)).to(dev)

    # per-launch events through a shim around the C entry point
    L = _lib.lib()
    real = L.sf_conv2d_f32
    recs = []

    class Shim:
        def __getattr__(self, n):
            return getattr(L, n)

        def sf_conv2d_f32(self, *a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = real(*a)
            e1.record()
            recs.append((a[5:14], e0, e1))
            return rc

    shim = Shim()
    old = _lib.lib
    with torch.no_grad():
        for _ in range(3):
            neck(second(x))
        torch.cuda.synchronize()
        _lib.lib = lambda: shim
        try:
            for _ in range(args.iters):
                neck(second(x))
            torch.cuda.synchronize()
        finally:
            _lib.lib = old
        shapes = {}
        for key, e0, e1 in recs:
            d = shapes.setdefault(key, [0.0, 0])
            d[0] += e0.elapsed_time(e1)
            d[1] += 1
        rows = []
        for (B, cin, h, w, cout, k, s, p, up), (ms, n) in shapes.items():
            ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
            fl = 2.0 * B * ho * wo * cout * up * up * cin * k * k
            us = ms / n * 1e3
            per_pass = n // args.iters
            rows.append(dict(cin=cin, cout=cout, h=h, w=w, k=k, stride=s, up=up, launches_per_pass=per_pass,
                             us=round(us, 2), gflop=round(fl / 1e9, 3),
                             tflops=round(fl / us / 1e6, 2), frac=round(fl / us / 1e6 / PEAK, 4)))
            print(f'{cin:4d}->{cout:4d} {h}x{w} k{k} s{s} up{up} x{per_pass}: {us:8.1f} us  '
                  f'{fl / 1e9:7.2f} GFLOP  {fl / us / 1e6:6.1f} TFLOP/s ({fl / us / 1e6 / PEAK:.1%})')
        # whole backbone + neck: eager and graph
        def run():
            return neck(second(x))[0]
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(args.iters):
            run()
        t1.record()
        torch.cuda.synchronize()
        eager_ms = t0.elapsed_time(t1) / args.iters
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            run()
            st.synchronize()
            with torch.cuda.graph(g, stream=st):
                run()
        torch.cuda.synchronize()
        t0.record()
        for _ in range(args.iters):
            g.replay()
        t1.record()
        torch.cuda.synchronize()
        graph_ms = t0.elapsed_time(t1) / args.iters
    tot_fl = sum(r['gflop'] * r['launches_per_pass'] for r in rows)
    print(f'SECOND+SECONDFPN batch {args.batch}: {tot_fl:.1f} GFLOP, eager {eager_ms:.3f} ms, '
          f'graph {graph_ms:.3f} ms = {tot_fl / graph_ms:.1f} TFLOP/s ({tot_fl / graph_ms / PEAK:.1%})')
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(dict(batch=args.batch, layers=rows, gflop=tot_fl, eager_ms=eager_ms,
                           graph_ms=graph_ms), f, indent=1)


if __name__ == '__main__':
    main()

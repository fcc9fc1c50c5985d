#!/usr/bin/env python3
"""The three segments of a frame (front = voxelizer + hash + rulebooks + plans | narrow layers |
wide layers + dense map), each replayed ALONE from its captured graph: mean ms per segment over
the distinct frames.  Under `rocprofv3 --kernel-trace --stats` the same run gives the per-kernel
durations with the GPU to each kernel alone.

    python tools/seg_bench.py [--iters 30] [--segments front,narrow,wide] [--dtype f32]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sparsefusion_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--distinct', type=int, default=4)
    ap.add_argument('--segments', default='front,narrow,wide')
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--strided-growth', type=float, default=None)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    B = args.batch
    host = [synth.nuscenes_frame(i) for i in range(args.distinct * B)]
    max_pts = max(p.shape[0] for p in host)
    cfg, ecfg, sd, enc, bb = bench.build_backbone('nusc', dev, B, max_pts, dtype=args.dtype,
                                                  strided_growth=args.strided_growth)
    frames = [[torch.from_numpy(host[f * B + b]).to(dev) for b in range(B)] for f in range(args.distinct)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    names = ['front', 'narrow', 'wide']
    want = args.segments.split(',')
    with torch.no_grad():
        sfs = [bb.capture_split(f, streams) for f in frames]
    torch.cuda.synchronize()
    tot = {}
    for si, name in enumerate(names):
        if name not in want:
            continue
        ms = []
        for sf in sfs:
            # the segments before this one must have run on THIS frame (buffers of the object)
            for sj in range(si):
                with torch.cuda.stream(streams[sj]):
                    sf.graphs[sj].replay()
                torch.cuda.synchronize()
            st = streams[si]
            with torch.cuda.stream(st):
                for _ in range(3):
                    sf.graphs[si].replay()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(args.iters):
                    sf.graphs[si].replay()
                e1.record(st)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / args.iters)
        tot[name] = float(np.mean(ms))
        print(f'{name:7s} alone: {tot[name]:.4f} ms per frame   (per frame: ' +
              ' '.join(f'{m:.3f}' for m in ms) + ')', flush=True)
    print('sum', round(sum(tot.values()), 4))
    if bb.results()['overflow']:
        print('OVERFLOW flag set')


if __name__ == '__main__':
    main()

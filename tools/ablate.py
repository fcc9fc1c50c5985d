#!/usr/bin/env python3
"""Which kernels bound the frame rate?  Times the bench loop with groups of launches left out of
the captured graph (the buffers keep the contents of the eager warm-up frame, so the remaining
kernels see realistic inputs; RESULTS ARE WRONG by construction -- this is a timing experiment).

    python tools/ablate.py [--streams 2] [--steps 120] none vox coord narrow c32 c64 c128 ...

Groups: vox (voxelizer), coord (hash + rulebooks except the first SubM table), coord0 (all of them),
narrow (convs with cout <= 32 and cin <= 16), c32 (32->32), c64 (cin or cout 64, cout <= 64),
c128 (cout 128), dense.  Several groups joined by '+' are left out together.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sparsefusion_amd import sharding, synth  # noqa: E402
from sparsefusion_amd.pipeline import LidarBackbone  # noqa: E402


class NoZero:
    def __init__(self, t):
        self.t = t

    def zero_(self):
        return self.t

    def __getattr__(self, n):
        return getattr(self.t, n)


class Shim:
    def __init__(self, real, skip):
        self.__dict__['real'] = real
        self.__dict__['skip'] = skip
        self.__dict__['nt'] = 0

    def __getattr__(self, name):
        f = getattr(self.real, name)
        skip = self.skip

        def group(cin, cout):
            return ('narrow' if cin <= 16 else 'c32' if (cin, cout) == (32, 32) else
                    'c64' if cout <= 64 else 'c128')

        def conv(*a):
            # sf_sparse_conv_f32(features_in, n_in, cin_padded, nbr, stride, K, k_first, w, cout, ...)
            return 0 if group(a[2], a[8]) in skip else f(*a)

        def conv_ordered(*a):
            # sf_sparse_conv_f32_ordered(features_in, n_in, cin_padded, nbr, stride, K, k_first, order, w, cout, ...)
            return 0 if group(a[2], a[9]) in skip else f(*a)

        def table(*a):
            self.__dict__['nt'] += 1
            first = self.nt == 1
            if 'coord0' in skip or ('coord' in skip and not first):
                return 0
            return f(*a)

        if name == 'sf_sparse_conv_f32':
            return conv
        if name == 'sf_sparse_conv_f32_ordered':
            return conv_ordered
        if name == 'sf_neighbor_table':
            return table
        if name in ('sf_strided_rulebook', 'sf_row_order_by_mask') and ('coord' in skip or 'coord0' in skip):
            return lambda *a: 0
        if name == 'sf_row_order_by_mask' and 'sort' in skip:
            return lambda *a: 0
        if name == 'sf_hash_build' and 'coord0' in skip:
            return lambda *a: 0
        if name == 'sf_hard_voxelize' and 'vox' in skip:
            return lambda *a: 0
        if name == 'sf_dense_bev' and 'dense' in skip:
            return lambda *a: 0
        return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--streams', type=int, default=2)
    ap.add_argument('--steps', type=int, default=120)
    ap.add_argument('--frames', type=int, default=4)
    ap.add_argument('groups', nargs='*', default=['none'])
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    host = [synth.nuscenes_frame(i) for i in range(args.frames)]
    max_pts = max(p.shape[0] for p in host)
    frames = [[torch.from_numpy(p).to(dev)] for p in host]
    for spec in args.groups:
        skip = set(spec.split('+')) - {'none'}
        cfg, ecfg, sd, enc, bb = bench.build_backbone('nusc', dev, 1, max_pts)
        lanes = [bb] + [LidarBackbone(cfg['pts_voxel_layer'], enc, dev, max_pts, batch_size=1)
                        for _ in range(args.streams - 1)]
        streams = [torch.cuda.Stream(device=dev) for _ in lanes]
        steps = []
        with torch.no_grad():
            for lane, st in zip(lanes, streams):
                with torch.cuda.stream(st):
                    fns = []
                    for f in frames:
                        lane.enqueue(f)           # real frame: every buffer valid
                        torch.cuda.synchronize()
                        real_L, real_counts = lane.L, lane.counts
                        lane.L = Shim(real_L, skip)
                        if 'vox' in skip:
                            lane.counts = NoZero(real_counts)
                        # capture() runs one eager enqueue itself, then captures: both see the shim
                        g = lane.capture(f)
                        lane.L, lane.counts = real_L, real_counts
                        fns.append(g.replay)
                steps.append((st, fns))
            torch.cuda.synchronize()

            def run(i):
                st, fns = steps[i % len(steps)]
                with torch.cuda.stream(st):
                    fns[(i // len(steps)) % len(fns)]()

            for i in range(16):
                run(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                run(i)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        print(f'{spec:24s} {el / args.steps * 1e3:7.3f} ms/frame  {args.steps / el:7.1f} frames/s', flush=True)
        del steps, lanes, bb


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Does the frame rate depend on WHERE the buffers lie?  Builds the two-lane bench set-up several
times in one process (the caching allocator hands out different addresses as blocks are freed and
reused), times it, and prints the addresses of the large buffers next to the rate.

    python tools/layout_probe.py [--builds 12] [--steps 160] [--hold N]

--hold N keeps N dummy allocations of odd sizes alive between builds to move the addresses.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sparsefusion_amd import synth  # noqa: E402
from sparsefusion_amd.pipeline import LidarBackbone  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--builds', type=int, default=12)
    ap.add_argument('--steps', type=int, default=160)
    ap.add_argument('--frames', type=int, default=4)
    ap.add_argument('--hold', type=int, default=0)
    ap.add_argument('--single', action='store_true', help='also time lane 0 alone')
    ap.add_argument('--no-overlap', action='store_true', help='no forked coordinate stream')
    ap.add_argument('--burn', type=int, default=0, help='extra streams created before each build')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    host = [synth.nuscenes_frame(i) for i in range(args.frames)]
    max_pts = max(p.shape[0] for p in host)
    frames = [[torch.from_numpy(p).to(dev)] for p in host]
    held = []
    for b in range(args.builds):
        burnt = [torch.cuda.Stream(device=dev) for _ in range(args.burn)]
        ov = not args.no_overlap
        cfg, ecfg, sd, enc, bb = bench.build_backbone('nusc', dev, 1, max_pts, overlap=ov)
        lanes = [bb, LidarBackbone(cfg['pts_voxel_layer'], enc, dev, max_pts, batch_size=1, overlap=ov)]
        streams = [torch.cuda.Stream(device=dev) for _ in lanes]
        steps = []
        with torch.no_grad():
            for lane, st in zip(lanes, streams):
                with torch.cuda.stream(st):
                    fns = [lane.capture(f).replay for f in frames]
                steps.append((st, fns))
            torch.cuda.synchronize()

            def run(i, nl=len(steps)):
                st, fns = steps[i % nl]
                with torch.cuda.stream(st):
                    fns[(i // nl) % len(fns)]()

            def timeit(nl):
                for i in range(16):
                    run(i, nl)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    run(i, nl)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / args.steps * 1e3

            w0 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
            t2 = timeit(2)
            w1 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
            print(f'window {b} {w0} {w1} {time.clock_gettime_ns(time.CLOCK_BOOTTIME) - w1} {time.time_ns() - w1}', flush=True)
            t1 = timeit(1) if args.single else float('nan')
        ptrs = []
        for lane in lanes:
            big = sorted(lane._tensors(), key=lambda t: -t.numel() * t.element_size())[:6]
            ptrs.append(' '.join(f'{t.data_ptr() >> 21:x}.{(t.data_ptr() >> 12) & 511:03x}' for t in big))
        print(f'build {b:2d}  2 lanes {t2:6.3f} ms ({1e3 / t2:6.1f} fps)  1 lane {t1:6.3f} ms | '
              f'reserved {torch.cuda.memory_reserved() >> 20} MiB', flush=True)
        print(f'          lane0 {ptrs[0]}', flush=True)
        print(f'          lane1 {ptrs[1]}', flush=True)
        del steps, lanes, bb, fns
        if args.hold:
            held.append(torch.empty((3 << 20) + 4096 * (b + 1), dtype=torch.uint8, device=dev))
            if len(held) > args.hold:
                held.pop(0)


if __name__ == '__main__':
    main()

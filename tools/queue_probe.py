#!/usr/bin/env python3
"""Which streams share a hardware queue, and which layout of a frame's segments over streams gives
the best frame rate?  (GPU_MAX_HW_QUEUES = 4 on the box: HIP maps every stream onto one of four
HSA queues; two streams on the same queue run their kernels one after the other.)

    python tools/queue_probe.py [--pairs] [--layouts] [--lanes 6] [--steps 160]

--pairs    one segment of frame A on one stream beside one segment of frame B on another (separate
           LidarBackbone objects, no ordering, every kernel does its real work): how much of the
           latency-bound front / narrow layers hides behind the matrix-bound wide layers
--layouts  LidarBackbone.capture_split with several (lanes, streams per segment, cut) layouts,
           ordered by events as in bench.py; prints ms per frame of each

Segments: front = voxelizer + hash + rulebooks + mask orders, narrow = layers in front of the cut
(default: the first layer with >= 64 input channels), wide = the rest + dense map.
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
from sparsefusion_amd.streams import queue_classes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pool', type=int, default=10)
    ap.add_argument('--steps', type=int, default=160)
    ap.add_argument('--frames', type=int, default=4)
    ap.add_argument('--lanes', type=int, default=4)
    ap.add_argument('--pairs', action='store_true')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    ap.add_argument('--layouts', action='store_true')
    ap.add_argument('--hide', action='store_true', help='light segments beside the wide layers: hidden share')
    ap.add_argument('--tail', action='store_true', help='with --layouts: only the variants that move the tail of a frame')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    host = [synth.nuscenes_frame(i) for i in range(args.frames)]
    max_pts = max(p.shape[0] for p in host)
    frames = [[torch.from_numpy(p).to(dev)] for p in host]
    cfg, ecfg, sd, enc, bb = bench.build_backbone('nusc', dev, 1, max_pts, True, args.dtype)
    lanes = [bb] + [LidarBackbone(cfg['pts_voxel_layer'], enc, dev, max_pts, batch_size=1, dtype=args.dtype)
                    for _ in range(max(args.lanes, 2) - 1)]
    pool = [torch.cuda.Stream(device=dev) for _ in range(args.pool)]
    cls = queue_classes(pool)
    print(f'streams that serialise (share a hardware queue): {cls}', flush=True)
    q = [pool[c[0]] for c in cls]
    Q = lambda i: q[i % len(q)]  # noqa: E731
    names = ['front', 'narrow', 'wide']

    with torch.no_grad():
        if args.pairs:
            fa = lanes[0].capture_split(frames[0], [Q(0), Q(1), Q(2)])
            fb = lanes[1].capture_split(frames[1], [Q(0), Q(1), Q(2)])
            torch.cuda.synchronize()

            def pair(sa, sb, n=args.steps // 2):
                def go():
                    for _ in range(n):
                        for seg in sa:
                            with torch.cuda.stream(Q(0)):
                                fa.graphs[seg].replay()
                        for seg in sb:
                            with torch.cuda.stream(Q(1)):
                                fb.graphs[seg].replay()
                go()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                go()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e3

            for sa, sb in [((0,), ()), ((1,), ()), ((2,), ()), ((1, 2), ()), ((0, 1, 2), ()),
                           ((0,), (2,)), ((0,), (1,)), ((1,), (2,)), ((0,), (1, 2)), ((0, 1), (2,)),
                           ((0, 1, 2), (0, 1, 2)), ((2,), (2,)), ((1,), (1,)), ((0,), (0,))]:
                t = pair(sa, sb)
                print(f'frame 0: {"+".join(names[i] for i in sa) or "-":20s} beside frame 1: '
                      f'{"+".join(names[i] for i in sb) or "-":20s} {t:6.3f} ms per round', flush=True)
            del fa, fb

        if args.hide:
            # how much of each light segment hides behind the wide layers of another frame
            cuts = [6, 10, 11]
            segn = ['front', 'layers 0-5 (<=16 ch)', 'layers 6-9 (32->32)', 'layer 10 (32->64)', 'wide']
            fa = lanes[0].capture_split(frames[1], [Q(0), Q(1), Q(2), Q(3), Q(0)], cut=cuts)
            fb = lanes[1].capture_split(frames[2], [Q(0), Q(1), Q(2), Q(3), Q(0)], cut=cuts)
            torch.cuda.synchronize()

            def run_pair(sa, sb, n=80):
                def go():
                    for _ in range(n):
                        if sa is not None:
                            with torch.cuda.stream(Q(0)):
                                fa.graphs[sa].replay()
                        if sb is not None:
                            with torch.cuda.stream(Q(1)):
                                fb.graphs[sb].replay()
                go()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                go()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e3
            tw = run_pair(None, 4)
            print(f'wide layers alone {tw:.3f} ms', flush=True)
            for sidx in range(4):
                ts = run_pair(sidx, None)
                tp = run_pair(sidx, 4)
                print(f'{segn[sidx]:24s} alone {ts:.3f} ms, beside the wide layers of another frame {tp:.3f} ms: '
                      f'{(ts + tw - tp) / ts * 100:5.1f} % hidden', flush=True)
            del fa, fb

        if args.layouts:
            def timeit(nl, assign, cut):
                steps = []
                for li in range(nl):
                    al = assign[li] if isinstance(assign[0], tuple) else assign
                    steps.append([lanes[li].capture_split(f, list(al), cut=cut) for f in frames])
                torch.cuda.synchronize()

                def run(i):
                    fr = steps[i % nl]
                    fr[(i // nl) % len(fr)].replay()
                for i in range(16):
                    run(i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    run(i)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / args.steps * 1e3

            def parity(nl, tail, k=2):
                """front + narrow of frame i on light stream i % k, then `tail`"""
                return tuple((Q(i % k), Q(i % k)) + tail for i in range(nl))

            tried = [
                ('everything on one stream', 2, (Q(0), Q(0), Q(0)), None),
                ('front | all layers', 2, (Q(0), Q(1), Q(1)), None),
                ('front | narrow | wide', 2, (Q(0), Q(1), Q(2)), None),
                ('front | narrow | wide', 3, (Q(0), Q(1), Q(2)), None),
                ('front | narrow | wide', 4, (Q(0), Q(1), Q(2)), None),
                ('front+narrow | wide', 2, (Q(0), Q(0), Q(1)), None),
                ('front+narrow by frame parity | wide', 3, parity(3, (Q(3),)), None),
                ('front+narrow by frame parity | wide', 4, parity(4, (Q(3),)), None),
                ('front+narrow by frame parity | wide', 6, parity(6, (Q(3),)), None),
                ('front+narrow on 3 streams | wide', 6, parity(6, (Q(3),), 3), None),
                ('same, cut in front of 32->64', 4, parity(4, (Q(3),)), 10),
                ('same, cut behind the first 64->64', 4, parity(4, (Q(3),)), 12),
                ('same, 64- and 128-channel layers apart', 4, parity(4, (Q(3), Q(2))), [11, 16]),
                ('front by frame parity | all layers', 4, tuple((Q(i % 2), Q(3), Q(3)) for i in range(4)), None),
                ('front by frame parity | narrow | wide', 4, tuple((Q(i % 2), Q(2), Q(3)) for i in range(4)), None),
                ('front by frame parity | narrow | wide', 6, tuple((Q(i % 2), Q(2), Q(3)) for i in range(6)), None),
                ('front on 3 streams | all layers', 6, tuple((Q(i % 3), Q(3), Q(3)) for i in range(6)), None),
                ('front+narrow by parity | wide by parity', 4, tuple((Q(i % 2), Q(i % 2), Q(2 + i % 2)) for i in range(4)), None),
                ('front+narrow by parity | wide by parity', 6, tuple((Q(i % 2), Q(i % 2), Q(2 + i % 2)) for i in range(6)), None),
                ('front+narrow by parity | wide by parity', 2, tuple((Q(i % 2), Q(i % 2), Q(2 + i % 2)) for i in range(2)), None),
                ('front+narrow by parity | wide, cut 6', 4, parity(4, (Q(3),)), 6),
                ('front+narrow by parity | wide, cut 16', 4, parity(4, (Q(3),)), 16),
            ]
            def par_tail(nl, cutlist):
                # the tail of every frame on the fourth queue (on a light stream it would queue up
                # behind the front of a LATER frame: 2.27 ms)
                return tuple((Q(i % 2), Q(i % 2), Q(3)) + (Q(2),) * (len(cutlist) - 1) for i in range(nl))
            if args.tail:
                tried = [('front+narrow by frame parity | wide', 4, parity(4, (Q(3),)), None),
                         ('same, dense map on the fourth queue', 4, par_tail(4, [11, 21]), [11, 21]),
                         ('same, conv_out + dense map on the fourth queue', 4, par_tail(4, [11, 20]), [11, 20]),
                         ('same, dense map on the fourth queue', 5, par_tail(5, [11, 21]), [11, 21]),
                         ('same, conv_out + dense map on the fourth queue', 5, par_tail(5, [11, 20]), [11, 20]),
                         ('front+narrow by frame parity | wide', 4, parity(4, (Q(3),)), None)]
            for name, nl, a, cut in tried:
                if nl > len(lanes):
                    continue
                t = timeit(nl, a, cut)
                print(f'{name:42s} {nl} frames in flight, cut {cut}:  {t:6.3f} ms/frame  '
                      f'{1e3 / t:6.1f} frames/s', flush=True)


if __name__ == '__main__':
    main()

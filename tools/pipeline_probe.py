#!/usr/bin/env python3
"""Where does a frame of bench.py's split pipeline wait?  Rebuilds the same pipeline (4 objects in
flight, front + narrow layers on one of two streams by frame parity, wide layers of every frame on
one stream) and records, per frame, the HOST time of every call of `_SplitFrame.replay` and the
GPU time of the segment boundaries (timing events against one base event).

    python tools/pipeline_probe.py [--frames 40] [--lanes 4]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sparsefusion_amd import synth  # noqa: E402
from sparsefusion_amd.pipeline import LidarBackbone  # noqa: E402
from sparsefusion_amd.streams import concurrent_streams  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=40)
    ap.add_argument('--lanes', type=int, default=4)
    ap.add_argument('--distinct', type=int, default=4)
    ap.add_argument('--wide-priority', type=int, default=None,
                    help='create the wide-layer stream with this priority (-1 = high) instead of '
                         'taking it from the probed pool')
    ap.add_argument('--quiet', action='store_true', help='totals only')
    ap.add_argument('--wide-streams', type=int, default=1,
                    help='2: the wide layers of even / odd frames on two streams (the tail of one frame\'s kernel is '
                         'filled by the other frame\'s)')
    ap.add_argument('--wg-times', action='store_true',
                    help='capture with sf_debug_dma_wg_times set: after the run, how level the workgroups of the LAST '
                         '128->128 launch finished INSIDE the running pipeline')
    ap.add_argument('--light-cus', type=int, default=0,
                    help='create the two light streams with a CU mask of this many CUs (hipExtStreamCreateWithCUMask)')
    ap.add_argument('--wide-rest', action='store_true', help='with --light-cus: the wide stream on the other CUs only')
    ap.add_argument('--ordered-grid-cap', type=int, default=0,
                    help='sf_debug_ordered_grid_cap while capturing: the persistent grid of the mask-ordered kernels')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    host = [synth.nuscenes_frame(i) for i in range(args.distinct)]
    max_pts = max(p.shape[0] for p in host)
    cfg, ecfg, sd, enc, bb = bench.build_backbone('nusc', dev, 1, max_pts)
    lanes = [bb] + [LidarBackbone(cfg['pts_voxel_layer'], enc, dev, max_pts) for _ in range(args.lanes - 1)]
    if args.light_cus:
        import ctypes
        hip = ctypes.CDLL('libamdhip64.so')
        words = 8

        def masked(bits):
            m = (ctypes.c_uint32 * words)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(words)])
            st = ctypes.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), words, m)
            assert rc == 0, rc
            return torch.cuda.ExternalStream(st.value, device=dev)
        lo = (1 << args.light_cus) - 1
        hi = ((1 << 256) - 1) ^ lo
        l0, l1, wide = masked(lo), masked(lo), masked(hi if args.wide_rest else (1 << 256) - 1)
        nq = 3
        print('CU masks: light', args.light_cus, 'CUs, wide', 'the rest' if args.wide_rest else 'all')
    elif args.wide_streams == 2:
        (l0, l1, wide, wide1), nq = concurrent_streams(dev, 4)
    elif args.wide_priority is None:
        (l0, l1, wide), nq = concurrent_streams(dev, 3)
    else:
        (l0, l1), nq = concurrent_streams(dev, 2)
        wide = torch.cuda.Stream(device=dev, priority=args.wide_priority)
        print('wide stream priority', wide.priority, 'range', torch.cuda.Stream.priority_range())
    print('hardware queues found', nq, flush=True)
    frames = [[torch.from_numpy(p).to(dev)] for p in host]
    wg_buf = None
    if args.wg_times:
        import ctypes
        from sparsefusion_amd import _lib
        raw = ctypes.CDLL(_lib.LIB_PATH)
        raw.sf_debug_dma_wg_times.argtypes = [ctypes.c_void_p]
        # one stamp buffer per captured graph (lane x frame): a replay overwrites the stamps of the graph's
        # earlier replays, the last 128->128 launch of the graph wins -- most graphs are replayed last in
        # mid-pipeline, only the final few while the pipeline drains
        wg_buf = torch.zeros(args.lanes * args.distinct, 2 * 4096, dtype=torch.int64, device=dev)
    if args.ordered_grid_cap:
        import ctypes
        from sparsefusion_amd import _lib
        raw_cap = ctypes.CDLL(_lib.LIB_PATH)
        raw_cap.sf_debug_ordered_grid_cap.argtypes = [ctypes.c_int]
        raw_cap.sf_debug_ordered_grid_cap(args.ordered_grid_cap)
    sfs = []
    with torch.no_grad():
        for li, lane in enumerate(lanes):
            light = (l0, l1)[li % 2]
            w_ = wide1 if (args.wide_streams == 2 and li % 2) else wide
            row = []
            for fi, f in enumerate(frames):
                if wg_buf is not None:
                    raw.sf_debug_dma_wg_times(wg_buf[li * args.distinct + fi].data_ptr())
                row.append(lane.capture_split(f, [light, light, w_]))
            sfs.append(row)
    if wg_buf is not None:
        raw.sf_debug_dma_wg_times(None)
    torch.cuda.synchronize()

    base = torch.cuda.Event(enable_timing=True)
    recs = []

    def replay(sf, rec):
        """_SplitFrame.replay with host stamps and timing events."""
        o = sf.owner
        prev = o._last_segment_done
        for i, (g, st, ev) in enumerate(zip(sf.graphs, sf.streams, sf.events)):
            h0 = time.perf_counter()
            if prev is not None:
                st.wait_event(prev)
            h1 = time.perf_counter()
            with torch.cuda.stream(st):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(st)
                h2 = time.perf_counter()
                g.replay()
                h3 = time.perf_counter()
                e1.record(st)
                ev.record(st)
            h4 = time.perf_counter()
            rec.append((i, h0, h1, h2, h3, h4, e0, e1))
            prev = ev
        o._last_segment_done = prev

    for rnd in range(2):       # round 0 warms up
        recs.clear()
        torch.cuda.synchronize()
        base.record()
        t0 = time.perf_counter()
        for i in range(args.frames):
            rec = []
            replay(sfs[i % len(lanes)][(i // len(lanes)) % len(frames)], rec)
            recs.append(rec)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f'round {rnd}: host submit {1e3 * (t1 - t0):.2f} ms, all done {1e3 * (t2 - t0):.2f} ms '
              f'= {1e3 * (t2 - t0) / args.frames:.3f} ms/frame')
    if wg_buf is not None:
        g = 256
        for gi, v in enumerate(wg_buf.cpu().numpy()):
            st, en = v[:g].astype(np.float64), v[g:2 * g].astype(np.float64)
            t0 = st.min()
            endr, busy, startr = (en - t0) / 100.0, (en - st) / 100.0, (st - t0) / 100.0
            print(f'lane {gi // args.distinct} frame {gi % args.distinct}: last 128->128 launch of the graph: span {endr.max():6.1f} us | '
                  f'starts p50 {np.percentile(startr, 50):5.1f} p90 {np.percentile(startr, 90):5.1f} max {startr.max():5.1f} | '
                  f'ends p10 {np.percentile(endr, 10):6.1f} p50 {np.percentile(endr, 50):6.1f} max {endr.max():6.1f} | '
                  f'busy mean {busy.mean():6.1f} max {busy.max():6.1f} us = {busy.mean() / endr.max():.1%} of the span')
    if args.quiet:
        return
    print('frame: host submit [ms from t0] (wait_event, replay calls) | light seg GPU start..end | wide seg GPU start..end')
    for i, rec in enumerate(recs):
        parts = []
        for (si, h0, h1, h2, h3, h4, e0, e1) in rec:
            parts.append(f'seg{si} host {1e3 * (h0 - t0):7.2f} +wait {1e3 * (h1 - h0):5.2f} +replay {1e3 * (h3 - h2):5.2f} '
                         f'| gpu {base.elapsed_time(e0):7.2f}..{base.elapsed_time(e1):7.2f}')
        print(f'{i:3d} ' + '  ||  '.join(parts))


if __name__ == '__main__':
    main()

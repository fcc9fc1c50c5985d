#!/usr/bin/env python3
"""Correctness of the split pipeline at full size: nuScenes-shaped frames in flight on 4
LidarBackbone objects (bench.py's layout), then every object's last frame compared -- rulebook and
BEV sha256 -- with the plain eager enqueue of the same frame on a separate object.

    python tools/split_check.py [--steps 64] [--lanes 4]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sparsefusion_amd import synth  # noqa: E402
from sparsefusion_amd.pipeline import LidarBackbone  # noqa: E402
from sparsefusion_amd.streams import concurrent_streams  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=64)
    ap.add_argument('--lanes', type=int, default=4)
    ap.add_argument('--frames', type=int, default=5)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    host = [synth.nuscenes_frame(i) for i in range(args.frames)]
    max_pts = max(p.shape[0] for p in host)
    frames = [[torch.from_numpy(p).to(dev)] for p in host]
    cfg, ecfg, sd, enc, ref = bench.build_backbone('nusc', dev, 1, max_pts)
    with torch.no_grad():
        want = []
        for f in frames:
            ref.enqueue(f)
            torch.cuda.synchronize()
            want.append(bench.frame_digest(ref))
        print('eager digests done', flush=True)
        lanes = [LidarBackbone(cfg['pts_voxel_layer'], enc, dev, max_pts, batch_size=1)
                 for _ in range(args.lanes)]
        (l0, l1, wide), nq = concurrent_streams(dev, 3)
        print('hardware queues', nq, flush=True)
        # every lane captures every frame: frame j of step i = (i // lanes + i) % frames varies per visit
        caps = [[lane.capture_split(f, [(l0, l1)[li % 2]] * 2 + [wide]) for f in frames]
                for li, lane in enumerate(lanes)]
        torch.cuda.synchronize()
        print('captured', flush=True)
        bad = 0

        def state(bb):
            torch.cuda.synchronize()
            c = bb.counts.tolist()
            out = {'counts': c, 'mean': bb.mean[:c[0]].clone(), 'num_points': bb.num_points[:c[0]].clone()}
            for (d, cout, dt), bufs in bb.fbufs.items():
                for j, t in enumerate(bufs):
                    out[f'set{d}_c{cout}_buf{j}'] = t[:c[d]].clone()
            for ri, rb in enumerate(bb.rulebooks):
                if 'order' in rb:
                    out[f'plan{ri}_order'] = rb['order'][:c[rb['dst']]].clone()
            return out

        def compare(a, b):
            for k in a:
                if k == 'counts':
                    same = a[k] == b[k]
                else:
                    same = a[k].shape == b[k].shape and bool(torch.equal(a[k], b[k]))
                if not same:
                    extra = ''
                    if k != 'counts' and a[k].shape == b[k].shape and a[k].dtype == torch.float32:
                        d = (a[k] != b[k])
                        rows = d.any(1).nonzero().flatten()
                        extra = f' rows differing {rows.numel()} of {a[k].shape[0]}, first {rows[:4].tolist()}, last {rows[-2:].tolist()}'
                    print('    DIFFERS:', k, extra, flush=True)

        for fi in (0, 1):
            ref.enqueue(frames[fi])
            want_state = state(ref)
            caps[2][fi].replay()
            got_state = state(lanes[2])
            print(f'buffers after frame {fi} on a fresh object (graph replay vs eager):', flush=True)
            compare(want_state, got_state)
        # 1. one object, one frame at a time, host sync after each: graph replay vs eager, no overlap
        seq = [0, 1, 0, 2, 4, 3, 3, 1]
        for k, fi in enumerate(seq):
            caps[0][fi].replay()
            torch.cuda.synchronize()
            got = bench.frame_digest(lanes[0])
            print(f'synced step {k} lane 0 frame {fi}: rulebooks {"same" if got[0] == want[fi][0] else "DIFFER"}, '
                  f'bev {"same" if got[1] == want[fi][1] else "DIFFERS"}', flush=True)
            bad += got != want[fi]
        # 2. the same without the sync, digest only at the end
        for fi in seq:
            caps[1][fi].replay()
        torch.cuda.synchronize()
        got = bench.frame_digest(lanes[1])
        print(f'one object, 8 frames back to back, last frame {seq[-1]}: rulebooks '
              f'{"same" if got[0] == want[seq[-1]][0] else "DIFFER"}, bev '
              f'{"same" if got[1] == want[seq[-1]][1] else "DIFFERS"}', flush=True)
        bad += got != want[seq[-1]]
        for rnd in range(3):
            last = {}
            for i in range(args.steps):
                li = i % args.lanes
                fi = (i * 3 + rnd) % args.frames
                caps[li][fi].replay()
                last[li] = fi
            torch.cuda.synchronize()
            for li, fi in sorted(last.items()):
                got = bench.frame_digest(lanes[li])
                ok = got == want[fi]
                bad += not ok
                print(f'round {rnd} lane {li} frame {fi}: rulebooks {"same" if got[0] == want[fi][0] else "DIFFER"}, '
                      f'bev {"same" if got[1] == want[fi][1] else "DIFFERS"}', flush=True)
        print('RESULT', 'all identical' if bad == 0 else f'{bad} mismatches', flush=True)
        sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()

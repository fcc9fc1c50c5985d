#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of tools/layout_probe.py per timed window: which HSA queue
ran which kernel group, and how much the groups of different queues overlapped in time.

    python tools/queue_trace.py <kernel_trace.csv> <layout_probe log>
"""
import csv
import sys
from collections import defaultdict


def group(name):
    if 'conv' in name:
        return 'conv'
    if name.startswith('(anonymous namespace)::vox') or 'vox_' in name:
        return 'vox'
    if 'dense' in name:
        return 'dense'
    return 'coord'


def main():
    trace, log = sys.argv[1], sys.argv[2]
    wins = []
    for line in open(log):
        if line.startswith('window '):
            f = line.split()
            wins.append((int(f[2]), int(f[3]), int(f[4]), int(f[5])))
    rows = []
    with open(trace) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'],
                         r.get('Stream_Id', '?'), r['Kernel_Name']))
    rows.sort()
    print(f'{len(rows)} kernels, first start {rows[0][0]}, windows {len(wins)}')
    # which clock?  try monotonic, boottime, realtime offsets
    for bi, (w0, w1, dboot, dreal) in enumerate(wins):
        sel = []
        for off in (0, dboot, dreal):
            if rows[0][0] <= w0 + off and w1 + off <= rows[-1][1] + 10**9:
                sel = [r for r in rows if w0 + off <= r[0] <= w1 + off]
                break
        if not sel:
            print(f'build {bi}: no kernels in window')
            continue
        span = max(r[1] for r in sel) - min(r[0] for r in sel)
        perq = defaultdict(lambda: defaultdict(lambda: [0, 0]))
        for s, e, q, st, n in sel:
            c = perq[(q, st)][group(n)]
            c[0] += 1
            c[1] += e - s
        # time with >= 2 conv kernels running
        ev = []
        for s, e, q, st, n in sel:
            if group(n) == 'conv':
                ev.append((s, 1))
                ev.append((e, -1))
        ev.sort()
        depth, last, both, anyc = 0, None, 0, 0
        for t, d in ev:
            if last is not None:
                if depth >= 2:
                    both += t - last
                if depth >= 1:
                    anyc += t - last
            depth += d
            last = t
        print(f'build {bi}: span {span / 1e6:.1f} ms, {len(sel)} kernels, conv busy {anyc / span:.2f}, '
              f'two convs at once {both / span:.2f}')
        for (q, st), g in sorted(perq.items()):
            print('    queue', q, 'stream', st, ' '.join(f'{k}:{v[0]}/{v[1] / 1e6:.1f}ms' for k, v in sorted(g.items())))


if __name__ == '__main__' and len(sys.argv) == 3:
    main()


def timeline(trace, log, build, t_from_ms, t_len_ms):
    wins = []
    for line in open(log):
        if line.startswith('window '):
            f = line.split()
            wins.append((int(f[2]), int(f[3])))
    w0, w1 = wins[build]
    rows = []
    with open(trace) as fh:
        for r in csv.DictReader(fh):
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            if w0 <= s <= w1:
                rows.append((s, e, r['Queue_Id'], r['Stream_Id'], r['Kernel_Name']))
    rows.sort()
    t0 = rows[0][0] + int(t_from_ms * 1e6)
    for s, e, q, st, n in rows:
        if t0 <= s <= t0 + int(t_len_ms * 1e6):
            short = n.replace('(anonymous namespace)::', '').replace('void ', '')
            short = short.split('(')[0][:44]
            print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q} s{st:>3s}  {short}')


if __name__ == '__main__' and len(sys.argv) > 3:
    timeline(sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5]))


def gaps(trace):
    """Time between consecutive kernels of the queue that runs the wide layers (bench.py's split
    pipeline), and what the other queues did meanwhile."""
    rows = []
    with open(trace) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']))
    rows.sort()
    perq = defaultdict(list)
    for r in rows:
        perq[r[2]].append(r)
    wq = max(perq, key=lambda q: sum(1 for r in perq[q] if 'conv_f32_dma_kernel<128' in r[3]))
    w = perq[wq]
    # steady state: the middle half of the wide queue's kernels
    w = w[len(w) // 4: 3 * len(w) // 4]
    others = [r for r in rows if r[2] != wq and w[0][0] <= r[0] <= w[-1][1]]
    import bisect
    ostarts = [r[0] for r in others]
    g, busy = [], 0
    for a, b in zip(w, w[1:]):
        gap = b[0] - a[1]
        busy += a[1] - a[0]
        lo, hi = bisect.bisect_left(ostarts, a[1]), bisect.bisect_right(ostarts, b[0])
        g.append((gap, hi - lo, a[3], b[3]))
    span = w[-1][1] - w[0][0]
    gs = sorted(x[0] for x in g)
    print(f'wide queue {wq}: {len(w)} kernels, busy {busy / span:.2f} of the span, gaps: mean {sum(gs) / len(gs) / 1e3:.1f} us, '
          f'median {gs[len(gs) // 2] / 1e3:.1f}, p90 {gs[int(len(gs) * .9)] / 1e3:.1f}, max {gs[-1] / 1e3:.1f}')
    by = defaultdict(list)
    for gap, n, a, b in g:
        short = lambda n_: n_.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:34]  # noqa: E731
        by[(short(a), short(b))].append((gap, n))
    for k, v in sorted(by.items(), key=lambda kv: -sum(x[0] for x in kv[1]))[:14]:
        print(f'  {k[0]:34s} -> {k[1]:34s} n={len(v):4d} mean gap {sum(x[0] for x in v) / len(v) / 1e3:7.1f} us, '
              f'kernels started on other queues meanwhile {sum(x[1] for x in v) / len(v):.1f}')


if __name__ == '__main__' and len(sys.argv) == 2:
    gaps(sys.argv[1])

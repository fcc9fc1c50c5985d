"""Streams that really run beside each other.

HIP maps every stream of a process onto one of GPU_MAX_HW_QUEUES hardware queues (4 on MI355X
unless the environment says otherwise), and two streams that share a queue execute their kernels
one after the other however independent they are.  Which queue a new stream gets depends on what
else the process has created, so the only reliable way to know is to look: two spin kernels on two
streams take as long as one if the streams have queues of their own and twice as long if they
share one.  `concurrent_streams` returns streams from pairwise different queues.
(the frame rate of the same captured frames varied by 20 % with this mapping.)"""
import time

import torch


def _spin_seconds(streams, cycles):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def queue_classes(streams, cycles=4_000_000):
    """Partition `streams` into classes that serialise with each other (= share a hardware queue)."""
    _spin_seconds(streams[:1], cycles)                      # warm-up (module load, clocks)
    one = min(_spin_seconds([s], cycles) for s in streams[:2] * 2)
    classes = []
    for i, s in enumerate(streams):
        for c in classes:
            both = min(_spin_seconds([streams[c[0]], s], cycles) for _ in range(2))
            if both > 1.6 * one:
                c.append(i)
                break
        else:
            classes.append([i])
    return classes


def concurrent_streams(device, n, candidates=12):
    """n DISTINCT streams of `device`, on pairwise different hardware queues as far as the probe
    finds that many (fewer: the rest are further streams of the pool, which then share a queue
    with one of the others -- the code runs, with less overlap; a probe disturbed by another
    process on the same GPU errs towards this side).  Returns (streams, distinct queues found)."""
    device = torch.device(device)
    with torch.cuda.device(device):
        pool = [torch.cuda.Stream(device=device) for _ in range(max(candidates, n))]
        classes = queue_classes(pool)
    picked = [pool[c[0]] for c in classes][:n]
    found = len(picked)
    depth = 1
    while len(picked) < n:          # second, third ... member of every class
        more = [pool[c[depth]] for c in classes if len(c) > depth]
        if not more:
            raise RuntimeError(f'{len(pool)} candidate streams cannot give {n} distinct ones')
        picked += more[:n - len(picked)]
        depth += 1
    return picked, found

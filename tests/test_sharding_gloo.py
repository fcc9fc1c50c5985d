"""N>1 path on CPU: two gloo ranks shard a frame stream round-robin, exchange only scalars and
checksums, and together cover every frame exactly once (no data-path collective)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sparsefusion_amd import sharding


def _worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        mine = sharding.frames_for_rank(rank, world, n_frames)
        # stand-in for the per-frame result: a deterministic function of the frame id
        local = {f: sharding.checksum(np.full(8, f, np.float32)) for f in mine}
        merged = sharding.gather_results(local)
        t = sharding.max_over_ranks(1.0 + rank)
        # the cross-check bench.py runs at N>1: the next rank recomputes frame 0 and every rank
        # verifies the merged table (complete, and the recomputed digest equal to the owner's)
        recheck = {('recheck', 0): sharding.checksum(np.full(8, 0, np.float32))} if rank == 1 % world else {}
        extra = sharding.gather_results(recheck)
        ok = sharding.verify_frames(merged, range(n_frames), [(0, extra[('recheck', 0)])])
        bad = None
        try:
            sharding.verify_frames(merged, range(n_frames + 1))
        except RuntimeError as e:
            bad = str(e)
        try:
            sharding.verify_frames(merged, range(n_frames), [(0, 'deadbeef')])
            bad2 = None
        except RuntimeError as e:
            bad2 = str(e)
        assert ok == n_frames and 'missing' in bad and 'recomputed' in bad2
        q.put((rank, mine, sorted(merged), t, merged[n_frames - 1]))
    finally:
        dist.destroy_process_group()


def test_two_rank_frame_sharding():
    world, n_frames = 2, 7
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]
    for r in res:
        assert r[2] == list(range(n_frames))          # every frame exactly once, on every rank
        assert r[3] == 2.0                            # max over ranks
        assert r[4] == sharding.checksum(np.full(8, n_frames - 1, np.float32))


def test_single_process_fallbacks():
    assert sharding.frames_for_rank(0, 1, 3) == [0, 1, 2]
    assert sharding.max_over_ranks(3.5) == 3.5
    assert sharding.gather_results({1: 'a'}) == {1: 'a'}

"""Multi-GPU driver pieces: the hot path shards by frame, one process per GPU, with NO collective
on the data path (frames are independent: hash tables, rulebooks and eval-mode BN are per frame;
SURVEY 8e; the reference's analogue is DistributedSampler + multi_gpu_test, tools/test.py:186-191).
Only scalars cross ranks, outside the timed region: the elapsed time (max) and result checksums."""
import hashlib

import numpy as np
import torch
import torch.distributed as dist


def frames_for_rank(rank, world, n_frames):
    """Round-robin: frame i of a stream goes to rank i % world."""
    return list(range(rank, n_frames, world))


def max_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    if dist.get_backend() != 'nccl':
        device = None          # gloo (CPU tests, rehearsals) reduces host tensors
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def checksum(array):
    """sha256 of a host array: what ranks exchange instead of 33 MB BEV maps."""
    return hashlib.sha256(np.ascontiguousarray(array).tobytes()).hexdigest()


def gather_results(local):
    """All ranks' {frame_id: checksum} dicts merged on every rank (a few KB)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(local)
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, dict(local))
    merged = {}
    for d in out:
        merged.update(d)
    return merged


def verify_frames(merged, expected_ids, rechecks=()):
    """Fail loudly when the job did not produce what a single process would have:
    every frame id of `expected_ids` must be present in `merged` (the gather_results dict), and every
    (frame_id, digest) in `rechecks` -- a frame recomputed on ANOTHER rank -- must equal the owning
    rank's digest.  A sharded run that silently dropped or corrupted a frame is not a result."""
    missing = sorted(i for i in expected_ids if i not in merged)
    if missing:
        raise RuntimeError(f'frames missing from the gathered results: {missing}')
    for fid, digest in rechecks:
        if merged[fid] != digest:
            raise RuntimeError(f'frame {fid}: digest {merged[fid]} on its rank, {digest} when another '
                               f'rank recomputed it')
    return len(expected_ids)

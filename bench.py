#!/usr/bin/env python3
"""Headline benchmark: LiDAR frames/s through voxelize + HardSimpleVFE mean + SparseEncoder
forward (BASELINE.json metric) on synthetic nuScenes-shaped sweeps, fp32, one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one frame (BASELINE.json configs[1]: single nuScenes-shaped frame, fp32) through the
whole hot path, points already resident in HBM, replayed from a HIP graph.  Frames shard by
rank (round-robin over a stream of frames), no collective on the data path; only the
elapsed time is max-reduced over ranks.  Rank 0 prints ONE JSON line.

Also reported on that line:
  roofline     -- the conv instantiation that takes the most time per frame: algorithmic
                  flops (2*P*Cin*Cout) or bytes (P*Cin*4 + N_out*Cout*4*(1+res) + P*8) per
                  launch / its mean launch duration measured with HIP events on the launch
                  stream; peak from MI355X_MICROARCH.md (fp32 MFMA 157.3 TFLOP/s, HBM 8 TB/s).
  cpu_baseline -- the CPU oracle (a port of the reference's CPU path, oracle/) on the same
                  frame 0, timed on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sparsefusion_amd import configs, sharding, synth  # noqa: E402
from sparsefusion_amd.pipeline import LidarBackbone  # noqa: E402
from sparsefusion_amd.sparse_encoder import SparseEncoder  # noqa: E402

PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA dense peak (no 2:1 sparsity)


def build_backbone(workload, dev, batch, max_points, overlap=True, dtype='f32', second=False,
                   strided_growth=None):
    cfg = configs.NUSC if workload == 'nusc' else configs.WAYMO
    ecfg = dict(cfg['pts_middle_encoder'])
    ecfg.pop('type')
    wcfg = {k: ecfg[k] for k in ('base_channels', 'output_channels', 'encoder_channels',
                                 'block_type') if k in ecfg}
    sd = synth.encoder_state_dict(2024, ecfg['in_channels'], **wcfg)
    enc = SparseEncoder(**ecfg)
    enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    enc = enc.to(dev).eval()
    vfe = None
    vcfg = dict(cfg['pts_voxel_encoder'])
    if vcfg.pop('type') == 'HardVFE':          # Waymo config: learned voxel encoder, seeded weights
        from sparsefusion_amd.voxel_encoder import HardVFE
        vfe = HardVFE(**vcfg)
        vsd = synth.hard_vfe_state_dict(2025, vfe.in_channels, vcfg['feat_channels'])
        vfe.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vsd.items()})
        vfe = vfe.to(dev).eval()
        sd = dict(sd, **{'vfe.' + k: v for k, v in vsd.items()})
    bev_backbone = neck = None
    if second:   # the BEV backbone and neck the map feeds (not part of the headline)
        from sparsefusion_amd.second import SECOND, SECONDFPN
        bcfg, ncfg = dict(cfg['pts_backbone']), dict(cfg['pts_neck'])
        bcfg.pop('type'), ncfg.pop('type')
        bev_backbone, neck = SECOND(**bcfg), SECONDFPN(**ncfg)
        bev_backbone.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in
                                      synth.second_state_dict(2026, **bcfg).items()})
        neck.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in
                              synth.second_fpn_state_dict(2027, **ncfg).items()})
        bev_backbone, neck = bev_backbone.to(dev).eval(), neck.to(dev).eval()
    bb = LidarBackbone(cfg['pts_voxel_layer'], enc, dev, max_points, batch_size=batch,
                       overlap=overlap, dtype=dtype, voxel_encoder=vfe, bev_backbone=bev_backbone,
                       neck=neck, strided_growth=strided_growth)
    return cfg, ecfg, sd, enc, bb


def layer_work(bb):
    """Per-layer algorithmic work of the LAST enqueued frame (SURVEY 8d / BASELINE.md sec. 4)."""
    counts = bb.counts.tolist()
    pairs = {}
    for i, rb in enumerate(bb.rulebooks):
        n = counts[rb['dst']]
        pairs[i] = int((rb['nbr'][:, :n] >= 0).sum().item())
    rows = []
    for lay in bb.layers:
        rb = lay.rb
        i = bb.rulebooks.index(rb)
        P, n_out = pairs[i], counts[rb['dst']]
        cin = lay.cin
        res = 1 if lay.residual_from == 'add' else 0
        si = 2.0 if lay.bf16 else 4.0                              # input / residual element bytes
        so = 2.0 if lay.out_dtype == torch.bfloat16 else 4.0       # output element bytes
        rows.append(dict(cin=cin, cout=lay.cout, subm=rb['subm'], K=rb['K'], pairs=P, n_out=n_out,
                         bf16=bool(lay.bf16), flops=2.0 * P * cin * lay.cout,
                         bytes=P * cin * si + n_out * lay.cout * (so + si * res) + P * 8.0))
    return rows, counts


def stage_work(bb, pts):
    """ALGORITHMIC bytes of the stages that are not convolutions, for the LAST enqueued batch
    (formulas of SURVEY 8d): voxelize = N F 4 read + M (max_pts F 4 + 12 + 4) write + M F 4 (fused
    mean); a rulebook = 16 bytes per input coordinate read + 8 bytes per pair written (+ 16 per new
    output coordinate of a strided rulebook); the 27 hash probes per row (8 bytes each) are listed
    apart; a plan (no reference counterpart) = its table read once + the permuted table, row order
    and headers written; dense = rows read once + the whole map written."""
    counts = bb.counts.tolist()
    n_pts = int(sum(p.shape[0] for p in pts))
    F = int(pts[0].shape[1])
    M = counts[0]
    out = {'voxelize': dict(points=n_pts, voxels=M,
                            bytes=n_pts * F * 4.0 + M * (bb.mp * F * 4.0 + 12 + 4) + M * F * 4.0)}
    si = 0
    for i, rb in enumerate(bb.rulebooks):
        n_in, n_out = counts[rb['src']], counts[rb['dst']]
        P = int((rb['nbr'][:, :n_out] >= 0).sum().item())
        if not rb['subm']:
            si += 1
        name = (f'subm_rulebook_set{rb["dst"]}' if rb['subm'] else f'strided_rulebook_{si}')
        out[name] = dict(rulebook=i, K=rb['K'], n_in=n_in, n_out=n_out, pairs=P,
                         bytes=n_in * 16.0 + P * 8.0 + (0 if rb['subm'] else n_out * 16.0),
                         probe_bytes=(rb['K'] * n_out * 8.0 if rb['subm'] else rb['K'] * n_in * 8.0))
        if 'order' in rb:
            out[f'plan_set{rb["dst"]}'] = dict(rulebook=i, rows=n_out,
                                              bytes=2.0 * rb['K'] * n_out * 4 + n_out * 4.0 +
                                              (n_out + 63) // 64 * 32 * 4.0)
    o = bb.sets[bb.out_set]
    D, H, W = o['shape']
    n_o = counts[bb.out_set]
    out['dense'] = dict(rows=n_o, bytes=n_o * bb.out_channels * 4.0 + n_o * 16.0 +
                        bb.B * bb.out_channels * D * H * W * 4.0)
    return out


def time_layers(bb, pts, vf, iters=5, busy=None):
    """Mean launch duration of every conv layer, HIP events on the launch stream (eager).

    Two things keep the event pairs honest (both measured against the rocprofv3 trace of the same
    run): (1) an eager enqueue is host bound (about a hundred ctypes calls per frame), so the
    stream is first parked on a calibrated spin kernel while the host queues the whole frame and
    the launches then run back to back; (2) an event is a marker with a release fence -- a pair
    around EVERY launch flushed the L2 between layers and read 220-233 us where the trace shows
    187 us -- so one pair brackets each RUN of consecutive layers of the same instantiation
    (cin, cout, dtype) and the run's time is divided by its launches (the four SubM layers and
    conv_out of the last stage are one run of five)."""
    import sparsefusion_amd._lib as _lib
    L = _lib.lib()
    keys = [(lay.cin, lay.cout, bool(lay.bf16), lay.rb['K'], lay.rb['subm']) for lay in bb.layers]
    nl = len(keys)
    first = [i == 0 or keys[i] != keys[i - 1] for i in range(nl)]
    last = [i == nl - 1 or keys[i] != keys[i + 1] for i in range(nl)]
    evs, state = [], {'i': 0}
    names = [None] * nl

    class Shim:
        def __getattr__(self, name):
            return getattr(L, name)

    def timed(real):
        def wrapped(*a):
            i = state['i']
            state['i'] = i + 1
            if first[i]:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
                evs.append([i, i, e0, None])
            rc = real(*a)
            names[i] = L.sf_last_conv_kernel().decode()
            if last[i]:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                evs[-1][1], evs[-1][3] = i, e1
            return rc
        return wrapped

    shim = Shim()
    shim.__dict__['sf_sparse_conv_f32'] = timed(L.sf_sparse_conv_f32)
    shim.__dict__['sf_sparse_conv_bf16'] = timed(L.sf_sparse_conv_bf16)
    shim.__dict__['sf_sparse_conv_f32_ordered'] = timed(L.sf_sparse_conv_f32_ordered)
    acc = np.zeros(nl)
    old = bb.L
    bb.L = shim
    try:
        for _ in range(iters):
            evs.clear()
            state['i'] = 0
            if busy is not None:
                busy()
            bb.enqueue(pts, vf)
            torch.cuda.synchronize()
            if state['i'] != nl:
                raise RuntimeError(f'{state["i"]} conv launches for {nl} layers')
            for a, b, ea, eb in evs:
                acc[a:b + 1] += ea.elapsed_time(eb) / (b - a + 1)
    finally:
        bb.L = old
    return acc / iters, names  # ms per layer (run-averaged), kernel launched per layer


def frame_digest(bb):
    """(rulebook sha256, BEV sha256) of the batch `bb` ran last.  The rulebook digest covers, for
    every rulebook of the encoder, the output coordinates and the neighbour table in the CPU
    reference's row order (first-touch order: the form the parity tests pin against the oracle),
    so it is the same on every GPU that ran the same frame; so is the BEV map (fixed summation
    order per output element)."""
    import hashlib
    torch.cuda.synchronize()
    c = bb.counts.tolist()
    h = hashlib.sha256()
    h.update(np.asarray(c[:-1], np.int64).tobytes())
    for rb in bb.rulebooks:
        n = c[rb['dst']]
        h.update(bb.sets[rb['dst']]['indices'][:n].cpu().numpy().tobytes())
        h.update(rb['nbr'][:, :n].contiguous().cpu().numpy().tobytes())
    return h.hexdigest(), sharding.checksum(bb.bev.cpu().numpy())


def measured_traffic(kernel, bf16=False):
    """HBM-side bytes per launch of `kernel` (the exact name rocprofv3 prints) from the committed
    PMC passes of this command (tools/pmc_traffic.sh: FETCH_SIZE / WRITE_SIZE collected and corrected
    as MI355X_MICROARCH.md prescribes).  Counters cannot be collected inside this process, so the
    number is read from the newest profiles/rNN_pmc_hbm_traffic*.json that lists the kernel; returns
    (bytes, 'profiles/<file>') or (None, None)."""
    import glob
    pat = 'r*_pmc_hbm_traffic_bf16.json' if bf16 else 'r*_pmc_hbm_traffic.json'
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', pat)), reverse=True):
        with open(path) as f:
            table = json.load(f)
        hits = [v for k, v in table.items() if k.replace(' ', '').startswith(kernel.replace(' ', ''))]
        if hits:
            n = sum(v['launches'] for v in hits)
            return (round(sum(v['traffic_bytes'] * v['launches'] for v in hits) / n, 1),
                    'profiles/' + os.path.basename(path))
    return None, None


def cpu_baseline(cfg, ecfg, sd, points, workload, repeats=5):
    """The oracle (port of the reference's CPU path) on one frame: one warm-up pass, then the
    median of `repeats` passes at the box's thread count, the per-stage split of SURVEY 8d
    (voxelize / VFE / rulebook / gather / GEMM / scatter-add / BN+ReLU / dense), and ONE pass on a
    single thread (`threads_1`, the reference's CPU ops are single-threaded apart from torch.mm).
    Returns (dict, bev of the last pass)."""
    from oracle import oracle as O
    O.lib()
    vl = cfg['pts_voxel_layer']
    mv = vl['max_voxels'][1] if isinstance(vl['max_voxels'], (tuple, list)) else vl['max_voxels']
    enc_sd = {k: v for k, v in sd.items() if not k.startswith('vfe.')}
    wcfg = {k: ecfg[k] for k in ('base_channels', 'output_channels', 'encoder_channels',
                                 'encoder_paddings', 'block_type') if k in ecfg}
    stage = {}

    def timed(name, fn):
        def wrapped(*a, **kw):
            t = time.perf_counter()
            r = fn(*a, **kw)
            stage[name] = stage.get(name, 0.0) + time.perf_counter() - t
            return r
        return wrapped

    def one_pass():
        stage.clear()
        O.conv_stage_times(reset=True)
        saved = (O.get_indice_pairs, O.bn_act, O.dense_bev, O.indice_conv)
        O.get_indice_pairs = timed('rulebook', saved[0])
        O.bn_act = timed('bn_relu', saved[1])
        O.dense_bev = timed('dense', saved[2])
        O.indice_conv = timed('conv', saved[3])
        try:
            t0 = time.perf_counter()
            v, c, n = O.hard_voxelize(points, vl['voxel_size'], vl['point_cloud_range'],
                                      vl['max_num_points'], mv)
            t1 = time.perf_counter()
            if workload == 'nusc':
                feats = O.vfe_mean(v, n, ecfg['in_channels'])
            else:
                vcfg = cfg['pts_voxel_encoder']
                layers = []
                for i in range(len(vcfg['feat_channels'])):
                    p = f'vfe.vfe_layers.{i}'
                    layers.append((sd[f'{p}.linear.weight'],
                                   {k: sd[f'{p}.norm.{k}'] for k in ('weight', 'bias', 'running_mean',
                                                                      'running_var')}))
                feats = O.hard_vfe(v, n, O.batch_coors([c]), layers, vcfg['voxel_size'],
                                   vcfg['point_cloud_range'], vcfg['norm_cfg']['eps'],
                                   vcfg['with_cluster_center'], vcfg['with_voxel_center'])
            t2 = time.perf_counter()
            bev = O.sparse_encoder_forward(enc_sd, feats, O.batch_coors([c]), 1, ecfg['sparse_shape'],
                                           ecfg['in_channels'], **wcfg)
            t3 = time.perf_counter()
        finally:
            O.get_indice_pairs, O.bn_act, O.dense_bev, O.indice_conv = saved
        g, m, sc = O.conv_stage_times(reset=True)
        split = dict(voxelize=t1 - t0, vfe=t2 - t1, rulebook=stage.get('rulebook', 0.0), gather=g,
                     gemm=m, scatter_add=sc, bn_relu=stage.get('bn_relu', 0.0),
                     dense=stage.get('dense', 0.0))
        split['other'] = (t3 - t0) - sum(split.values())
        return t3 - t0, split, bev, c.shape[0]

    threads = int(O.lib().sfo_num_threads())
    one_pass()                                             # warm-up: page in, allocator, clocks
    runs = [one_pass() for _ in range(repeats)]
    runs.sort(key=lambda r: r[0])
    secs, split, bev, nvox = runs[len(runs) // 2]
    O.lib().sfo_set_num_threads(1)
    try:
        secs1 = one_pass()[0]
    finally:
        O.lib().sfo_set_num_threads(threads)
    return dict(value=round(1.0 / secs, 4), unit='frames/s', cores=threads, kind='port',
                threads_1=round(1.0 / secs1, 4), passes=repeats,
                spread=[round(1.0 / runs[-1][0], 4), round(1.0 / runs[0][0], 4)],
                stages_ms={k: round(v * 1e3, 1) for k, v in split.items()},
                sample=f'1 full frame (frame 0, {points.shape[0]} points -> {nvox} voxels): 1 warm-up + '
                       f'median of {repeats} passes at {threads} threads, then 1 pass at 1 thread; '
                       f'OpenMP on the GEMM rows only, rest single-threaded as in the reference'), bev


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='nusc', choices=['nusc', 'waymo'])
    ap.add_argument('--frames', type=int, default=None,
                    help='distinct synthetic frames (batches) per rank; default 4 on one GPU, and '
                         '64 / (N * batch) on N > 1 GPUs: the 64-frame stream of BASELINE configs[3] '
                         'dealt round-robin over the ranks')
    ap.add_argument('--batch', type=int, default=1, help='frames per step (samples per batch)')
    ap.add_argument('--streams', type=int, default=None,
                    help='frames in flight per GPU, each with its own workspace (default 4 with the '
                         'split pipeline, 2 with --single-graph)')
    ap.add_argument('--single-graph', action='store_true',
                    help="one HIP graph per frame with the coordinate phase forked inside it, one stream "
                         "per frame in flight (round 1's loop; the hardware queues of the graph's "
                         "branches are then the runtime's choice) instead of the split pipeline")
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help="f32 = the headline config (BASELINE configs[1]); bf16 = configs[2] "
                         "(bf16 MFMA, fp32 accumulation; use with --batch 8)")
    ap.add_argument('--second', action='store_true',
                    help='also run SECOND + SECONDFPN on the BEV map inside the step (SURVEY 8f-3; '
                         'an extended workload, not the headline metric)')
    ap.add_argument('--strided-growth', type=float, default=None,
                    help='bound on output/input rows of a strided conv for sizing (default: the '
                         'geometric worst case, 8); overflow is detected, never silent')
    ap.add_argument('--frame-pipeline', action='store_true',
                    help='submit the frames through pipeline.FramePipeline (graphs captured once per '
                         'object, every frame COPIED into a static point buffer first: what a caller '
                         'with a stream of new frames runs) instead of graphs bound to the resident frames')
    ap.add_argument('--cut', type=int, default=None,
                    help='split pipeline: first layer of the wide-layer segment (default: the first '
                         'layer with >= 64 input channels)')
    ap.add_argument('--ordered', default=None,
                    help="A/B runs: fp32 layer shapes on the mask-ordered kernel ('all', 'none' or "
                         "a list like 32x32,64x64); default: spconv.ORDERED_SHAPES")
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-overlap', action='store_true',
                    help='coordinate phase on the main stream instead of a forked one')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-passes', type=int, default=5, help='timed passes of the CPU baseline')
    ap.add_argument('--detail', default='', help='write per-layer timing/roofline JSON here')
    args = ap.parse_args()
    if args.ordered is not None:
        from sparsefusion_amd import spconv as _spconv
        _spconv.ORDERED_SHAPES = args.ordered

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    # One rank per GPU.  SF_BENCH_BACKEND=gloo is a REHEARSAL switch for a box with fewer GPUs than
    # ranks (ranks then share devices, scalars cross ranks over gloo); the driver's runs use RCCL.
    backend = os.environ.get('SF_BENCH_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()
    if backend == 'nccl' and local >= ndev:
        raise SystemExit(f'LOCAL_RANK {local} but {ndev} GPU(s) visible')
    dev = torch.device('cuda', local % max(ndev, 1))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.frames is None:
        args.frames = 4 if world == 1 else max(1, 64 // (world * args.batch))
    split = not (args.single_graph or args.no_graph)
    if args.streams is None:
        args.streams = 4 if split else 2
    gen = synth.nuscenes_frame if args.workload == 'nusc' else synth.waymo_frame
    B = args.batch
    # weak scaling: a stream of world*frames*B frames dealt round-robin, rank r gets r, r+world, ...
    ids = sharding.frames_for_rank(rank, world, world * args.frames * B)
    host_frames = [gen(i) for i in ids]
    max_pts = max(p.shape[0] for p in host_frames)
    cfg, ecfg, sd, enc, bb = build_backbone(args.workload, dev, B, max_pts, not args.no_overlap,
                                            args.dtype, args.second, args.strided_growth)
    # frames in flight: independent frames on independent streams fill each other's tail rounds
    # and barrier bubbles; every stream has its own workspace, weights are shared
    lanes = [bb] + [LidarBackbone(cfg['pts_voxel_layer'], enc, dev, max_pts, batch_size=B,
                                  overlap=not args.no_overlap, dtype=args.dtype,
                                  voxel_encoder=bb.vfe, bev_backbone=bb.bev_backbone, neck=bb.neck,
                                  strided_growth=args.strided_growth)
                    for _ in range(args.streams - 1)]
    if split:
        # front + narrow layers of frame i on light[i % 2], the wide layers of every frame on one
        # stream; the three streams on hardware queues of their own (LidarBackbone.capture_split)
        from sparsefusion_amd.streams import concurrent_streams
        # bf16 (front-bound): the narrow layers on the fourth queue instead of behind their front
        # (1125 vs 1096 frames/s; fp32: 494 vs 496 -- profiles/r02_queue_probe*.txt)
        (light0, light1, wide_st, narrow_st), n_queues = concurrent_streams(dev, 4)
        if args.dtype != 'bf16':
            narrow_st = None
        streams = [wide_st for _ in lanes]
    else:
        n_queues, narrow_st = None, None
        streams = [torch.cuda.Stream(device=dev) for _ in lanes]
    frames = [[torch.from_numpy(host_frames[f * B + b]).to(dev) for b in range(B)]
              for f in range(args.frames)]
    vf = None
    torch.cuda.synchronize()

    split_frames = []  # split pipeline: per lane, the captured frames
    wide_events = []   # split pipeline: (start, end) events around the wide-layer graph of every frame
    with torch.no_grad():
        steps = []   # steps[i] = (stream, callable): frame i % frames on lane i % streams
        for li, (lane, st) in enumerate(zip(lanes, streams)):
            with torch.cuda.stream(st):
                if args.no_graph:
                    fns = [(lambda f=f, lane=lane: lane.enqueue(f, vf)) for f in frames]
                elif split:
                    light = (light0, light1)[li % 2]
                    sfs = [lane.capture_split(f, [light, narrow_st or light, wide_st], cut=args.cut,
                                              voxel_features=vf)
                           for f in frames]
                    split_frames.append(sfs)
                    fns = [(lambda sf=sf: sf.replay(timed=wide_events)) for sf in sfs]
                else:
                    fns = [lane.capture(f, vf).replay for f in frames]
            steps.append((st, fns))
        torch.cuda.synchronize()

        def run(i):
            st, fns = steps[i % len(steps)]
            with torch.cuda.stream(st):
                fns[(i // len(steps)) % len(fns)]()

        if args.frame_pipeline:
            if B != 1 or not split or args.second or args.workload != 'nusc':
                raise SystemExit('--frame-pipeline: nuScenes workload, one frame per step, split pipeline')
            from sparsefusion_amd.pipeline import FramePipeline
            pipe = FramePipeline(cfg['pts_voxel_layer'], enc, dev, max_pts, frames[0][0].shape[1],
                                 in_flight=args.streams, dtype=args.dtype,
                                 strided_growth=args.strided_growth)

            def run(i):   # noqa: F811
                pipe.submit(frames[i % len(frames)][0])
            for i in range(2 * args.streams):     # every object captures on its first frame
                run(i)

        for i in range(args.warmup):
            run(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        wide_events.clear()
        t0 = time.perf_counter()
        for i in range(args.steps):
            run(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        elapsed = sharding.max_over_ranks(elapsed, dev)   # the only cross-rank traffic

        # wide-layer graph: mean start-to-end inside the running pipeline (the events of the timed
        # region) and alone (one frame at a time on lane 0)
        wide_ms = None
        if split and rank == 0 and wide_events:
            in_pipe = float(np.mean([a.elapsed_time(b) for a, b in wide_events]))
            alone = []
            for _ in range(3):
                for sf in split_frames[0]:
                    sf.replay(timed=alone)
            torch.cuda.synchronize()
            alone_ms = float(np.mean([a.elapsed_time(b) for a, b in alone[len(steps[0][1]):]]))
            wide_ms = (in_pipe, alone_ms)
        wide_events.clear()

        # ---- roofline of the dominant conv kernel (rank 0), outside the timed region -------
        roofline, detail = None, None
        if rank == 0:
            # every distinct frame of the timed region contributes equally (the frames differ: the
            # last stage of frame 0 holds 613 k pairs, of frame 3 about 440 k), so that the average
            # launch duration is the one a rocprofv3 trace of this run shows
            nb = max(2, int(np.ceil(12.0 / max(elapsed / args.steps * 1e3, 1e-3))))   # ~12 ms

            def keep_busy(fi):
                """The same frame's graph replayed ahead of the timed eager frame on this stream:
                clocks up, host ahead of the GPU.  The other lanes stay idle: a launch is timed
                with the GPU to itself, which is also what a rocprofv3 kernel trace reports (the
                trace of this run agrees within 2 %; with the other lane replaying beside, the
                same launches read 230 us instead of 175 because they share the CUs)."""
                for _ in range(nb):
                    steps[0][1][fi]()
                if split:   # the replays run on the pipeline's streams: the timed frame follows them
                    torch.cuda.current_stream(dev).wait_event(bb._last_segment_done)

            ms, work, counts, knames = None, None, None, None
            for fi, fr in enumerate(frames):
                busy = None if args.no_graph else (lambda: keep_busy(fi))
                ms_f, knames = time_layers(bb, fr, vf, iters=3, busy=busy)
                bb.enqueue(fr, vf)
                torch.cuda.synchronize()
                work_f, counts_f = layer_work(bb)
                stage_f = stage_work(bb, fr)
                if fi == 0:
                    ms, work, counts, stages = ms_f, work_f, counts_f, stage_f
                else:
                    ms = ms + ms_f
                    for k_, st_ in stage_f.items():
                        for kk in st_:
                            if kk not in ('rulebook', 'K'):
                                stages[k_][kk] += st_[kk]
                    for w, wf in zip(work, work_f):
                        for k in ('pairs', 'n_out', 'flops', 'bytes'):
                            w[k] += wf[k]
            ms = ms / len(frames)
            for w in work:
                for k in ('pairs', 'n_out', 'flops', 'bytes'):
                    w[k] /= len(frames)
            for st_ in stages.values():
                for kk in st_:
                    if kk not in ('rulebook', 'K'):
                        st_[kk] /= len(frames)
            bb.enqueue(frames[0], vf)
            torch.cuda.synchronize()
            res = bb.results()
            if res['overflow']:
                raise SystemExit('overflow flag set: a bound was exceeded')
            groups = {}
            for w, t, kn in zip(work, ms, knames):
                gkey = (w['cin'], w['cout'], w['bf16'], w['K'], w['subm'])
                g_ = groups.setdefault(gkey, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0,
                                                  kernels=[]))
                w['kernel'] = kn
                if kn not in g_['kernels']:
                    g_['kernels'].append(kn)
                g_['ms'] += t
                g_['flops'] += w['flops']
                g_['bytes'] += w['bytes']
                g_['launches'] += 1
                w['ms'] = float(t)
            dom = max(groups, key=lambda k: groups[k]['ms'])
            gd = groups[dom]
            avg_s = gd['ms'] / gd['launches'] * 1e-3
            fl, by = gd['flops'] / gd['launches'], gd['bytes'] / gd['launches']
            peak_mfma = PEAK_BF16_MFMA_TFLOPS if dom[2] else PEAK_F32_MFMA_TFLOPS
            t_mfma, t_hbm = fl / (peak_mfma * 1e12), by / (PEAK_HBM_GBS * 1e9)
            if t_mfma >= t_hbm:
                ach = fl / avg_s / 1e12
                roofline = dict(bound='mfma', achieved=round(ach, 3), peak=peak_mfma,
                                unit='TFLOP/s', frac=round(ach / peak_mfma, 4),
                                traffic=None)
            else:
                ach = by / avg_s / 1e9
                roofline = dict(bound='hbm', achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                                frac=round(ach / PEAK_HBM_GBS, 4), traffic=None)
            # the kernel of the group's heaviest layer, named exactly as rocprofv3 prints it
            heavy = max((w for w in work if (w['cin'], w['cout'], w['bf16'], w['K'], w['subm']) == dom),
                        key=lambda w: w['ms'])
            roofline['kernel'] = heavy['kernel']
            roofline['kernels_in_group'] = gd['kernels']
            roofline['traffic'], roofline['traffic_source'] = measured_traffic(heavy['kernel'], dom[2])
            roofline['launches_per_frame'] = gd['launches']
            roofline['avg_launch_us'] = round(avg_s * 1e6, 2)
            # avg_launch_us (and so `achieved`) is measured with the GPU to itself, after the timed
            # region; `value` is measured in the running pipeline, where the same launches share the
            # CUs with the other streams' kernels and take longer (profiles/: rocprofv3 of this command)
            roofline['timing'] = 'alone'
            if wide_ms is not None:
                # for the record: start-to-end of the wide-layer graph in the timed region and alone.
                # The difference is time between its kernels (the queue waits while kernels of the
                # other streams hold the CUs), not longer launches: the rocprofv3 trace of this
                # command shows the launches of the timed region within 3 % of avg_launch_us
                roofline['wide_graph_ms'] = {'in_pipeline': round(wide_ms[0], 4), 'alone': round(wide_ms[1], 4)}
            roofline['algorithmic_flops_per_launch'] = fl
            roofline['algorithmic_bytes_per_launch'] = by
            tot_fl, tot_by = sum(w['flops'] for w in work), sum(w['bytes'] for w in work)
            detail = dict(counts=counts[:-1], layers=work, conv_ms_per_frame=float(ms.sum()),
                          frame_flops=tot_fl, frame_bytes=tot_by,
                          groups={f'{k[0]}x{k[1]}' + ('b' if k[2] else '') +
                                  ('' if k[3] == 27 else f'k{k[3]}') + ('' if k[4] else 's'): v
                                  for k, v in groups.items()},
                          stages=stages, workspace_bytes=bb.memory_bytes())

        # ---- what the job computed, checked across ranks (outside the timed region) -------------
        # every batch of every rank must be present in the gathered table, and a batch recomputed
        # on ANOTHER rank (rank 1 redoes rank 0's first batch) must give the same rulebooks and map
        local = {}
        for fi, fr in enumerate(frames):
            bb.enqueue(fr, vf)
            local[tuple(ids[fi * B:(fi + 1) * B])] = frame_digest(bb)
        merged = sharding.gather_results(local)
        recheck = {}
        if world > 1 and rank == 1:
            ids0 = sharding.frames_for_rank(0, world, world * args.frames * B)[:B]
            if max(gen(i).shape[0] for i in ids0) <= max_pts:
                bb.enqueue([torch.from_numpy(gen(i)).to(dev) for i in ids0], vf)
                recheck[('recheck',) + tuple(ids0)] = frame_digest(bb)
        recheck = sharding.gather_results(recheck)
        expected = []
        for r in range(world):
            ids_r = sharding.frames_for_rank(r, world, world * args.frames * B)
            expected += [tuple(ids_r[f * B:(f + 1) * B]) for f in range(args.frames)]
        verified = sharding.verify_frames(merged, expected,
                                          [(k[1:], v) for k, v in recheck.items()])

    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.steps * B / elapsed
    prec = 'fp32' if args.dtype == 'f32' else ('bf16 storage + bf16 MFMA with fp32 accumulation '
                                               '(first layer fp32)')
    if rank == 0:
        line = {
            'metric': 'LiDAR frames/sec (voxelize+SparseEncoder fwd)',
            'value': round(value, 2), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': ('nuScenes-shaped 10-sweep frame (sparsefusion_nusc_voxel_LC_r50: '
                                    f'0.075 m voxels, sparse_shape 41x1440x1440), {prec}, {B} frame(s) '
                                    'per step' if args.workload == 'nusc' else
                                    'Waymo-shaped stress frame (transfusion_waymo_voxel_L: 0.1 m '
                                    f'voxels, 41x1504x1504, HardVFE 5->64 + SparseEncoder), {prec}'),
                       'frames_per_step': B, 'distinct_frames_per_rank': args.frames,
                       'points_per_frame': int(np.mean([p.shape[0] for p in host_frames])),
                       'voxels_frame0': detail['counts'][0] if detail else None,
                       'bev_backbone': 'SECOND + SECONDFPN (fp32) on the BEV map' if args.second else None,
                       'hip_graph': not args.no_graph, 'frames_in_flight': args.streams,
                       'frame_pipeline': bool(args.frame_pipeline),
                       'pipeline': (('split: front on 2 streams by frame parity, narrow layers on 1, wide layers '
                                     if narrow_st is not None else
                                     'split: front + narrow layers on 2 streams by frame parity, wide layers ') +
                                    f'on 1, {n_queues} hardware queues found' if split else
                                    'one graph per frame, coordinate phase forked inside'), 'sharding': 'frames by rank, no collective'},
            'roofline': roofline,
            'verified': {'batches_checked': verified, 'rechecked_on_another_rank': len(recheck),
                         'frame0_rulebook_sha256': merged[expected[0]][0][:16],
                         'frame0_bev_sha256': merged[expected[0]][1][:16]},
        }
        if detail:
            line['frame_algorithmic_gflop'] = round(detail['frame_flops'] / 1e9, 2)
            line['frame_algorithmic_gbyte'] = round(detail['frame_bytes'] / 1e9, 3)
        if world == 1 and not args.no_cpu_baseline:
            cb, bev_cpu = cpu_baseline(cfg, ecfg, sd, host_frames[0], args.workload, args.cpu_passes)
            line['cpu_baseline'] = cb
            if B == 1 and args.dtype == 'f32':  # same frame, same math: the GPU result must match
                with torch.no_grad():
                    bb.enqueue(frames[0], vf)
                    got = bb.bev.cpu().numpy()
                err = float(np.abs(got - bev_cpu).max() / max(np.abs(bev_cpu).max(), 1e-30))
                line['parity_vs_cpu_rel_err'] = err
        else:
            line['cpu_baseline'] = None
        if args.detail and detail:
            with open(args.detail, 'w') as f:
                json.dump(detail, f, indent=1)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""Generate the committed golden fixtures (run in the build container, where /root/reference and
therefore oracle/_ref exist):

    python tests/golden/make_golden.py

voxelize_*.npz   inputs + outputs of the REFERENCE's own CPU voxelizer (oracle/_ref, compiled from
                 mmdet3d/ops/voxel/src by oracle/build_ref.py).  voxelize_kitti_like uses the first
                 4096 points of the reference's demo/kitti_000008.bin (a data file, not source)
                 with caps that both trigger.
encoder_*.npz    a small synthetic frame through the SparseEncoder wiring (rulebook
                 counts and sha256 per layer, per-layer feature statistics, the dense BEV) with the
                 REFERENCE's own spconv-1.x CPU code doing the work: every rulebook comes from the
                 compiled getIndicePairsSubM / getIndicePairsConv (include/spconv/geometry.h) and
                 every convolution from the compiled gather / scatter-add functors
                 (src/reordering.cc) + torch::mm_out, through oracle/_ref/spconv (built by
                 oracle/build_ref.py).  Only the layer wiring, eval-mode BatchNorm and the dense
                 scatter are the oracle's (oracle.sparse_encoder_forward with the two functions
                 swapped).  The oracle must reproduce them: rulebooks bit for bit, features to
                 1e-5 (GEMM summation order).
fullsize_rulebooks.json
                 SURVEY G5/G6: sha256 of the CPU-reference-ordered rulebooks (indiceNum, outids,
                 used pair prefixes), row / pair counts for all 8 encoder rulebooks of the
                 full-size frames the BASELINE configs name (nuScenes frame 0 at 41x1440x1440,
                 B = 8 nuScenes frames 0-7, Waymo-shaped frame 0 at 41x1504x1504 with the
                 150 000-voxel cap hit), computed BY THE COMPILED REFERENCE, plus the oracle's
                 per-layer feature abs-mean for nuScenes frame 0.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

from oracle import build_ref, oracle as O  # noqa: E402
from sparsefusion_amd import configs, synth  # noqa: E402
from helpers import SMALL_RANGE, SMALL_SHAPE, SMALL_VOXEL, small_points  # noqa: E402


def ref_voxelize(mod, pts, vs, rng, mp, mv):
    import torch
    p = torch.from_numpy(np.ascontiguousarray(pts, np.float32))
    vox = torch.zeros(mv, mp, p.shape[1])
    co = torch.zeros(mv, 3, dtype=torch.int32)
    npv = torch.zeros(mv, dtype=torch.int32)
    k = mod.hard_voxelize(p, vox, co, npv, [float(x) for x in vs], [float(x) for x in rng], mp, mv, 3)
    return vox[:k].numpy(), co[:k].numpy(), npv[:k].numpy()


class reference_spconv:
    """Context manager: oracle.get_indice_pairs / oracle.indice_conv -> the compiled reference."""

    def __init__(self, ref):
        self.ref = ref

    def __enter__(self):
        import torch
        ref = self.ref
        self.saved = (O.get_indice_pairs, O.indice_conv)

        def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0,
                             dilation=1, subm=False):
            ks, st, pd, dl = (O._triple(v) for v in (ksize, stride, padding, dilation))
            oshape = list(spatial_shape) if subm else O.conv_output_size(spatial_shape, ks, st, pd, dl)
            o, p, n = ref.get_indice_pairs(torch.from_numpy(np.ascontiguousarray(indices, np.int32)),
                                           int(batch_size), oshape, ks, st, pd, dl, bool(subm))
            return o.numpy().copy(), p.numpy(), n.numpy(), oshape

        def indice_conv(features, filters_kio, pairs, num, n_out, subm):
            t = lambda a, d: torch.from_numpy(np.ascontiguousarray(a, d))
            return ref.indice_conv(t(features, np.float32), t(filters_kio, np.float32),
                                   t(pairs, np.int32), t(num, np.int32), int(n_out), False,
                                   bool(subm)).numpy()

        O.get_indice_pairs, O.indice_conv = get_indice_pairs, indice_conv
        return self

    def __exit__(self, *exc):
        O.get_indice_pairs, O.indice_conv = self.saved


ENCODER_CHAIN = [(3, 1, 1, True), (3, 2, 1, False), (3, 1, 1, True), (3, 2, 1, False),
                 (3, 1, 1, True), (3, 2, (0, 1, 1), False), (3, 1, 1, True),
                 ((3, 1, 1), (2, 1, 1), 0, False)]


def rulebook_digest(outids, pairs, num):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(num, np.int32).tobytes())
    h.update(np.ascontiguousarray(outids, np.int32).tobytes())
    for k in range(pairs.shape[0]):
        h.update(np.ascontiguousarray(pairs[k, :, :int(num[k])], np.int32).tobytes())
    return h.hexdigest()


def fullsize(ref):
    """G5/G6: digests of the full-size rulebooks, from the compiled reference."""
    out = {}
    nus, way = configs.NUSC, configs.WAYMO

    def chain(idx, batch, shape):
        rows = []
        with reference_spconv(ref):
            for ks, st, pd, subm in ENCODER_CHAIN:
                o, p, n, oshape = O.get_indice_pairs(idx, batch, shape, ks, st, pd, 1, subm)
                rows.append(dict(n_in=int(idx.shape[0]), n_out=int(o.shape[0]), pairs=int(n.sum()),
                                 sha256=rulebook_digest(o, p, n)))
                idx, shape = o, oshape
        return rows

    def voxels(cfg, pts):
        vl = cfg['pts_voxel_layer']
        mv = vl['max_voxels']
        mv = mv[1] if isinstance(mv, (tuple, list)) else mv
        return O.hard_voxelize(pts, vl['voxel_size'], vl['point_cloud_range'],
                               vl['max_num_points'], mv)

    shape = list(nus['pts_middle_encoder']['sparse_shape'])
    v, c, n = voxels(nus, synth.nuscenes_frame(0))
    out['nusc_frame0'] = chain(O.batch_coors([c]), 1, shape)
    # per-layer feature statistics of the ORACLE on that frame (regression pin, not reference data)
    enc = {k: nus['pts_middle_encoder'][k] for k in ('output_channels', 'encoder_channels',
                                                     'encoder_paddings', 'block_type')}
    sd = synth.encoder_state_dict(1, 5, **{k: enc[k] for k in ('output_channels', 'encoder_channels',
                                                               'block_type')})
    trace = {}
    bev = O.sparse_encoder_forward(sd, O.vfe_mean(v, n, 5), O.batch_coors([c]), 1, shape, 5,
                                   trace=trace, **enc)
    out['nusc_frame0_features'] = dict(
        weight_seed=1, layers=[dict(name=k, abs_mean=float(np.abs(t['features']).mean()),
                                    abs_max=float(np.abs(t['features']).max()))
                               for k, t in trace.items() if isinstance(t, dict)],
        bev_abs_mean=float(np.abs(bev).mean()), bev_nonzero=int(np.count_nonzero(bev)))
    coors = [voxels(nus, synth.nuscenes_frame(i))[1] for i in range(8)]
    out['nusc_b8'] = chain(O.batch_coors(coors), 8, shape)
    v, c, n = voxels(way, synth.waymo_frame(0))
    out['waymo_frame0'] = chain(O.batch_coors([c]),
                                1, list(way['pts_middle_encoder']['sparse_shape']))
    out['waymo_frame0_voxels'] = int(c.shape[0])
    import json
    with open(os.path.join(HERE, 'fullsize_rulebooks.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('fullsize_rulebooks.json', {k: (len(v_) if isinstance(v_, list) else v_)
                                      for k, v_ in out.items() if k != 'nusc_frame0_features'})


def main():
    mod = build_ref.load_prebuilt() or build_ref.build()
    assert mod is not None, 'needs /root/reference'
    ref = build_ref.load_prebuilt_spconv() or build_ref.build_spconv()
    assert ref is not None, 'needs /root/reference and triton\'s cuda_runtime_api.h'
    kitti = np.fromfile('/root/reference/demo/kitti_000008.bin', dtype=np.float32).reshape(-1, 4)[:4096]
    pts = np.concatenate([kitti, np.zeros((kitti.shape[0], 1), np.float32)], 1)
    vs, rng = [0.075, 0.075, 0.2], [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
    v, c, n = ref_voxelize(mod, pts, vs, rng, 3, 512)
    np.savez_compressed(os.path.join(HERE, 'voxelize_kitti_like.npz'), points=pts,
                        voxel_size=np.float32(vs), coors_range=np.float32(rng), max_points=3,
                        max_voxels=512, voxels=v, coors=c, num_points=n)
    pts = small_points(99, 3000)
    pts[17, 0] = 8.0
    pts[18, 1] = -8.0
    v, c, n = ref_voxelize(mod, pts, SMALL_VOXEL, SMALL_RANGE, 4, 700)
    np.savez_compressed(os.path.join(HERE, 'voxelize_small_caps.npz'), points=pts,
                        voxel_size=np.float32(SMALL_VOXEL), coors_range=np.float32(SMALL_RANGE),
                        max_points=4, max_voxels=700, voxels=v, coors=c, num_points=n)

    cfg = dict(encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
               encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)),
               block_type='basicblock')
    for name, batch in (('encoder_b1', 1), ('encoder_b2', 2)):
        frames = [small_points(500 + b, 3000 - 1200 * b) for b in range(batch)]
        vox, coors = [], []
        for p in frames:
            v, c, n = O.hard_voxelize(p, SMALL_VOXEL, SMALL_RANGE, 10, 20000)
            vox.append(O.vfe_mean(v, n, 5))
            coors.append(c)
        feats, coors4 = np.concatenate(vox, 0), O.batch_coors(coors)
        sd = synth.encoder_state_dict(77, 5, encoder_channels=cfg['encoder_channels'],
                                      block_type='basicblock')
        trace = {}
        with reference_spconv(ref):
            bev = O.sparse_encoder_forward(sd, feats, coors4, batch, SMALL_SHAPE, 5, trace=trace,
                                           **cfg)
        layers = [k for k in trace if isinstance(trace[k], dict)]
        out = dict(points_seed=np.int64([500 + b for b in range(batch)]),
                   points_n=np.int64([3000 - 1200 * b for b in range(batch)]),
                   weight_seed=77, voxel_features=feats, coors=coors4,
                   bev_nonzero_idx=np.flatnonzero(bev).astype(np.int32),
                   bev_nonzero_val=bev.ravel()[np.flatnonzero(bev)], bev_shape=np.int64(bev.shape),
                   layer_names=np.array(layers),
                   layer_rows=np.int64([trace[k]['outids'].shape[0] for k in layers]),
                   layer_indice_num=np.stack([np.pad(trace[k]['num'], (0, 27 - len(trace[k]['num'])))
                                              for k in layers]).astype(np.int32),
                   layer_feat_abs_mean=np.float64([np.abs(trace[k]['features']).mean() for k in layers]),
                   layer_outids_sha=np.array([hashlib.sha256(np.ascontiguousarray(
                       trace[k]['outids']).tobytes()).hexdigest() for k in layers]),
                   layer_pairs_sha=np.array([hashlib.sha256(np.ascontiguousarray(
                       trace[k]['pairs']).tobytes()).hexdigest() for k in layers]))
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    fullsize(ref)
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == '__main__':
    main()

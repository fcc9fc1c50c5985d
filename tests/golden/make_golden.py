"""Generate the committed golden fixtures (run in the build container, where /root/reference and
therefore oracle/_ref exist):

    python tests/golden/make_golden.py

voxelize_*.npz   inputs + outputs of the REFERENCE's own CPU voxelizer (oracle/_ref, compiled from
                 mmdet3d/ops/voxel/src by oracle/build_ref.py).  voxelize_kitti_like uses the first
                 4096 points of the reference's demo/kitti_000008.bin (a data file, not source)
                 with caps that both trigger.
encoder_*.npz    a small synthetic frame through the CPU ORACLE's SparseEncoder wiring (SURVEY G3/G4:
                 rulebook counts per layer, per-layer feature statistics, the dense BEV).  The
                 reference's spconv cannot be built here, so these pin the oracle against
                 regressions and the GPU path against the oracle; they are not reference outputs.
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
from sparsefusion_amd import synth  # noqa: E402
from helpers import SMALL_RANGE, SMALL_SHAPE, SMALL_VOXEL, small_points  # noqa: E402


def ref_voxelize(mod, pts, vs, rng, mp, mv):
    import torch
    p = torch.from_numpy(np.ascontiguousarray(pts, np.float32))
    vox = torch.zeros(mv, mp, p.shape[1])
    co = torch.zeros(mv, 3, dtype=torch.int32)
    npv = torch.zeros(mv, dtype=torch.int32)
    k = mod.hard_voxelize(p, vox, co, npv, [float(x) for x in vs], [float(x) for x in rng], mp, mv, 3)
    return vox[:k].numpy(), co[:k].numpy(), npv[:k].numpy()


def main():
    mod = build_ref.load_prebuilt() or build_ref.build()
    assert mod is not None, 'needs /root/reference'
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
        bev = O.sparse_encoder_forward(sd, feats, coors4, batch, SMALL_SHAPE, 5, trace=trace, **cfg)
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
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == '__main__':
    main()

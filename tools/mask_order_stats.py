#!/usr/bin/env python3
"""CPU analysis (not product code): MFMA row-blocks issued / pairs for the SubM rulebooks of one
synthetic nuScenes frame under different row orders -- natural, global mask sort, and mask sort
WITHIN chunks of consecutive rows (keeps gather locality, one LDS-local sort per chunk).

    python tools/mask_order_stats.py [frame]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import oracle as O  # noqa: E402
from sparsefusion_amd import configs, synth  # noqa: E402
from helpers import ENCODER_CHAIN  # noqa: E402


def waste(masks, order, blk=16):
    m = masks[order]
    n = len(m)
    pad = (-n) % blk
    if pad:
        m = np.concatenate([m, np.zeros(pad, m.dtype)])
    m = m.reshape(-1, blk)
    ored = np.bitwise_or.reduce(m, axis=1)
    issued = sum(int(((ored >> k) & 1).sum()) for k in range(27)) * blk
    pairs = sum(int(((masks >> k) & 1).sum()) for k in range(27))
    return issued / pairs


def main():
    fi = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    cfg = configs.NUSC
    vl = cfg['pts_voxel_layer']
    pts = synth.nuscenes_frame(fi)
    v, c, n = O.hard_voxelize(pts, vl['voxel_size'], vl['point_cloud_range'], vl['max_num_points'],
                              vl['max_voxels'][1])
    ind = O.batch_coors([c])
    shape = list(cfg['pts_middle_encoder']['sparse_shape'])
    for (ks, st, pd, subm) in ENCODER_CHAIN:
        outids, pairs, num, oshape = O.get_indice_pairs(ind, 1, shape, ks, st, pd, 1, subm)
        if subm:
            nrow = ind.shape[0]
            masks = np.zeros(nrow, np.int64)
            for k in range(27):
                masks[pairs[k, 1, :num[k]]] |= (1 << k)
            nat = np.arange(nrow)
            line = [f'rows {nrow:7d} pairs/row {num.sum() / nrow:5.2f} | natural {waste(masks, nat):.3f}',
                    f'global {waste(masks, np.argsort(masks, kind="stable")):.3f}']
            for ch in (256, 512, 1024, 2048, 4096, 8192, 16384):
                order = np.concatenate([s + np.argsort(masks[s:s + ch], kind='stable')
                                        for s in range(0, nrow, ch)])
                line.append(f'c{ch} {waste(masks, order):.3f}')
            # 64-row skip (whole tile) for the chunk orders
            print('  '.join(line), flush=True)
        ind, shape = outids, oshape


if __name__ == '__main__':
    main()

"""Experiment: how much do the conv layers gain if voxel rows are spatially ordered?
Sorts the POINTS on the host by a block-major cell key, so the voxelizer's first-point order
becomes a spatial order, and times every conv layer (HIP events) for both orders."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from sparsefusion_amd import synth, configs

dev = torch.device('cuda:0')
pts = synth.nuscenes_frame(0)
vl = configs.NUSC['pts_voxel_layer']
vs, r = np.asarray(vl['voxel_size'], np.float32), np.asarray(vl['point_cloud_range'], np.float32)

def keyed(pts, mode):
    c = np.floor((pts[:, :3] - r[:3]) / vs).astype(np.int64)
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    if mode == 'raster':
        k = (z * 1440 + y) * 1440 + x
    elif mode.startswith('block'):
        b = int(mode[5:])
        k = (((y // b) * (1440 // b + 1) + x // b) * 64 + z) * (b * b) + (y % b) * b + (x % b)
    elif mode.startswith('morton'):
        b = int(mode[6:])
        by, bx = (y // b).clip(0, 1023), (x // b).clip(0, 1023)
        def spread(v):
            v = v.astype(np.int64)
            out = np.zeros_like(v)
            for i in range(10):
                out |= ((v >> i) & 1) << (2 * i)
            return out
        k = ((spread(by) << 1 | spread(bx)) * 64 + z) * (b * b) + (y % b) * b + (x % b)
    return np.ascontiguousarray(pts[np.argsort(k, kind='stable')])

cfg, ecfg, sd, enc, bb = bench.build_backbone('nusc', dev, 1, pts.shape[0])
for mode in ['orig'] + sys.argv[1:]:
    p = pts if mode == 'orig' else keyed(pts, mode)
    f = [torch.from_numpy(p).to(dev)]
    with torch.no_grad():
        bb.enqueue(f); torch.cuda.synchronize()
        ms = bench.time_layers(bb, f, None, iters=5)
        work, counts = bench.layer_work(bb)
    g = {}
    for w, t in zip(work, ms):
        g.setdefault((w['cin'], w['cout']), 0.0)
        g[(w['cin'], w['cout'])] += t
    print('%-9s counts %s conv %.3f ms | ' % (mode, counts[:5], ms.sum()) + '  '.join('%dx%d %.3f' % (k[0], k[1], v) for k, v in g.items()))

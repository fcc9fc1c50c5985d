import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from sparsefusion_amd import synth, _lib
dev = torch.device('cuda:0')
pts = synth.nuscenes_frame(0)
cfg, ecfg, sd, enc, bb = bench.build_backbone('nusc', dev, 1, pts.shape[0])
f = [torch.from_numpy(pts).to(dev)]
L = _lib.lib()

raw = ctypes.CDLL(_lib.LIB_PATH)
raw.sf_debug_conv_profile.argtypes = [ctypes.c_void_p]
buf = torch.zeros(16, dtype=torch.int64, device=dev)
with torch.no_grad():
    bb.enqueue(f); torch.cuda.synchronize()
    real = L.sf_sparse_conv_f32
    names = ['setup', 'issue', 'mma', 'bar1', 'store', 'bar2', 'epi', 'phases', 'cyc', 'ticks', 'wgs']
    class Shim:
        def __getattr__(self, n): return getattr(L, n)
    res = []
    def wrapped(*a):
        buf.zero_(); torch.cuda.synchronize()
        raw.sf_debug_conv_profile(buf.data_ptr())
        rc = real(*a); torch.cuda.synchronize()
        raw.sf_debug_conv_profile(None)
        v = buf.cpu().numpy().astype(np.float64)
        res.append((a[2], a[8], v.copy()))
        return rc
    sh = Shim(); sh.__dict__['sf_sparse_conv_f32'] = wrapped
    bb.L = sh
    bb.enqueue(f); torch.cuda.synchronize()
for cin, cout, v in res:
    wg = max(v[10], 1); cyc = v[8]
    clk = v[8] / max(v[9], 1) * 100  # MHz
    parts = v[:7] / cyc * 100
    print('%3dx%3d wgs=%5d clk=%4.0fMHz cyc/wg=%8.0f phases/wg=%5.1f | ' % (cin, cout, wg, clk, cyc / wg, v[7] / wg) +
          ' '.join('%s %4.1f%%' % (n, p) for n, p in zip(names, parts)) + ' | cyc/phase=%.0f' % (cyc / max(v[7], 1)))

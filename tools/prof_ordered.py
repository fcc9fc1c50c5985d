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
raw.sf_debug_ordered_profile.argtypes = [ctypes.c_void_p]
buf = torch.zeros(16, dtype=torch.int64, device=dev)
with torch.no_grad():
    bb.enqueue(f); torch.cuda.synchronize()
    real = L.sf_sparse_conv_f32_ordered
    class Shim:
        def __getattr__(self, n): return getattr(L, n)
    res = []
    def wrapped(*a):
        buf.zero_(); torch.cuda.synchronize()
        raw.sf_debug_ordered_profile(buf.data_ptr())
        rc = real(*a); torch.cuda.synchronize()
        raw.sf_debug_ordered_profile(None)
        res.append((a[2], a[9], buf.cpu().numpy().astype(np.float64)))
        return rc
    sh = Shim(); sh.__dict__['sf_sparse_conv_f32_ordered'] = wrapped
    bb.L = sh
    bb.enqueue(f); torch.cuda.synchronize()
for cin, cout, v in res:
    w = max(v[7], 1); ph = max(v[4], 1)
    print('%3dx%3d waves=%5d cyc/wave=%8.0f phases/wave=%5.1f (mult %.1f) | per phase: issue %5.0f  mma %5.0f  epi %5.0f  barrier %5.0f  sum %5.0f | clock %4.0f MHz' % (
        cin, cout, w, v[6] / w, v[4] / w, v[5] / w, v[0] / ph, v[1] / ph, v[2] / ph, v[3] / ph, v[:4].sum() / ph, v[6] / max(v[8], 1) * 100))

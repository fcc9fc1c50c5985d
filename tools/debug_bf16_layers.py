"""Per-layer comparison of the bf16 module path with the rounding-emulating oracle (diagnostic)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import SMALL_SHAPE, to_dev, weight_cfg
from test_gpu_conv import make_encoder
from test_gpu_bf16 import encoder_inputs, rms
from oracle import oracle as O
from sparsefusion_amd import spconv

dev = torch.device('cuda:0')
enc, sd, cfg = make_encoder(dev, 5, 21)
feats, coors4 = encoder_inputs(O, 1)
kw = dict(bn_eps=1e-3, encoder_paddings=cfg['encoder_paddings'], **weight_cfg(cfg))
tr16, tr32 = {}, {}
O.sparse_encoder_forward(sd, feats, coors4, 1, SMALL_SHAPE, 5, bf16=True, trace=tr16, **kw)
O.sparse_encoder_forward(sd, feats, coors4, 1, SMALL_SHAPE, 5, trace=tr32, **kw)
with torch.no_grad():
    x = spconv.SparseConvTensor(to_dev(feats, dev), to_dev(coors4, dev), SMALL_SHAPE, 1)
    x = enc.conv_input(x)
    def cmp(name, x):
        g = x.features.float().cpu().numpy(); e = tr16[name]['features']; f = tr32[name]['features']
        print('%-42s rows %6d  vs16 max %.2e rms %.2e | vs32 max %.2e rms %.2e' % (
            name, g.shape[0], np.abs(g - e).max() / np.abs(e).max(), rms(g - e) / rms(e),
            np.abs(g - f).max() / np.abs(f).max(), rms(g - f) / rms(f)))
    cmp('conv_input', x)
    x = x.replace_feature(spconv.cast_bf16(x.features))
    for i, stage in enumerate(enc.encoder_layers):
        for j, blk in enumerate(stage):
            x = blk(x)
            cmp(f'encoder_layers.encoder_layer{i + 1}.{j}', x)
    x = enc.conv_out(x, out_dtype=torch.float32)
    cmp('conv_out', x)

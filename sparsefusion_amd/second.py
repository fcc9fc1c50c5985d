"""`SECOND` and `SECONDFPN`: drop-ins for mmdet3d/models/backbones/second.py:8-86 and
mmdet3d/models/necks/second_fpn.py:11-92, the first consumers of the BEV map
`SparseEncoder` returns (detectors/sparsefusion.py:123-126).

Same constructor arguments, attributes and state-dict keys (`blocks.I.J.*`, `deblocks.I.J.*`: the
modules ARE nn.Sequential chains of nn.Conv2d / nn.ConvTranspose2d / nn.BatchNorm2d / nn.ReLU, used
as parameter containers).  In eval mode every conv + BN + ReLU triple is ONE launch of the
hand-written implicit-GEMM kernel (`sf_conv2d_f32`, csrc/conv2d_f32.hip) with the BatchNorm folded
into the epilogue; the neck's deblocks write straight into their channel slice of the concatenated
map, so `torch.cat` (second_fpn.py:88-89) disappears.  With training-mode BatchNorm the kernel
computes the bare convolution and torch applies BN / ReLU on the device.  No CPU path.
"""
import numpy as np
import torch
from torch import nn

from . import _lib
from .registry import NORM_LAYERS, Registry

BACKBONES = Registry('backbone')
NECKS = Registry('neck')

_SUPPORTED = {(3, 1), (3, 2), (1, 1), (2, 2)}


def _build_norm(norm_cfg, channels):
    cfg = dict(norm_cfg)
    cls = NORM_LAYERS.get(cfg.pop('type'))
    if cls is None or not issubclass(cls, nn.BatchNorm2d):
        raise NotImplementedError(f'norm layer {norm_cfg} (the target configs use BN)')
    cfg.pop('requires_grad', None)
    return cls(channels, **cfg)


def _build_conv(conv_cfg, cin, cout, k, **kw):
    cfg = dict(conv_cfg)
    t = cfg.pop('type', 'Conv2d')
    if t not in ('Conv2d', 'Conv', None):
        raise NotImplementedError(f'conv layer type {t}')
    return nn.Conv2d(cin, cout, k, **kw, **cfg)


def _build_upsample(upsample_cfg, in_channels, out_channels, kernel_size, stride):
    cfg = dict(upsample_cfg)
    t = cfg.pop('type')
    if t != 'deconv':
        raise NotImplementedError(f'upsample layer type {t} (the target configs use deconv)')
    return nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride=stride, **cfg)


def _fold_bn2d(bn, bias):
    """Eval BatchNorm2d (+ conv bias) -> (scale, shift); tools/misc/fuse_conv_bn.py:18-21."""
    scale = torch.rsqrt(bn.running_var.float() + bn.eps)
    if bn.weight is not None:
        scale = scale * bn.weight.float()
    shift = -bn.running_mean.float() * scale
    if bn.bias is not None:
        shift = shift + bn.bias.float()
    if bias is not None:
        shift = shift + bias.float() * scale
    return scale.contiguous(), shift.contiguous()


class _PackedFilters:
    """Filters of one layer in the kernel's fragment order, re-packed when the parameter changes."""

    def __init__(self):
        self.key, self.packed = None, None
        self.fold_key, self.fold_val = None, None

    def fold(self, conv, bn):
        """Folded eval-BatchNorm (scale, shift), recomputed only when a parameter or buffer changed
        (a handful of tiny torch kernels per layer otherwise end up in every captured graph)."""
        ts = [t for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var, conv.bias)
              if t is not None]
        key = tuple((t.data_ptr(), t._version) for t in ts) + (bn.eps,)
        if key != self.fold_key:
            self.fold_key, self.fold_val = key, _fold_bn2d(bn, conv.bias)
        return self.fold_val

    def get(self, conv):
        w = conv.weight
        key = (w.data_ptr(), w._version, w.device)
        if key == self.key:
            return self.packed
        _lib.require(w, torch.float32, 'conv weight')
        if isinstance(conv, nn.ConvTranspose2d):
            cin, cout, kh, kw = w.shape
            up, ks = kh, 1
            mat = w.detach().permute(2, 1, 3, 0).reshape(kh * kw * cout, cin).contiguous()   # (dy, co, dx)
        else:
            cout, cin, kh, kw = w.shape
            up, ks = 1, kh
            mat = w.detach().contiguous()
        L = _lib.lib()
        nbytes = L.sf_conv2d_packed_bytes(cout, cin, ks, up)
        packed = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
        _lib.check(L.sf_conv2d_pack_filters_f32(_lib.ptr(mat), cout, cin, ks, up, packed.data_ptr(),
                                                _lib.stream(w.device)))
        self.key, self.packed = key, packed
        return packed


def _check_conv(conv):
    one = lambda v: v[0] if isinstance(v, tuple) else v
    sq = lambda v: (not isinstance(v, tuple)) or all(x == v[0] for x in v)
    if not all(sq(v) for v in (conv.kernel_size, conv.stride, conv.padding, conv.dilation)):
        raise NotImplementedError('non-square conv2d parameters')
    if one(conv.dilation) != 1 or conv.groups != 1:
        raise NotImplementedError('dilation / groups are not used by SECOND / SECONDFPN')
    k, s, p = one(conv.kernel_size), one(conv.stride), one(conv.padding)
    if isinstance(conv, nn.ConvTranspose2d):
        if k != s or p != 0 or one(conv.output_padding) != 0:
            raise NotImplementedError('ConvTranspose2d needs kernel_size == stride, padding 0 '
                                      '(second_fpn.py:46-51 builds exactly that)')
        return 1, 1, 0, k
    if (k, s) not in _SUPPORTED:
        raise NotImplementedError(f'Conv2d kernel {k} / stride {s}; built: {sorted(_SUPPORTED)}')
    if getattr(conv, 'padding_mode', 'zeros') != 'zeros':
        raise NotImplementedError('padding_mode != zeros')
    return k, s, p, 1


def conv_bn_relu(x, conv, bn, relu, cache, out=None, channel_offset=0):
    """relu(bn(conv(x))) as one launch.  `out` (B, C_total, H', W') receives channels
    [channel_offset, channel_offset + Cout); allocated (Cout channels) when None."""
    _lib.require(x, torch.float32, 'x')
    if x.dim() != 4:
        raise RuntimeError(f'expected (N, C, H, W), got {tuple(x.shape)}')
    k, s, p, up = _check_conv(conv)
    cin = conv.in_channels
    cout = conv.out_channels
    B, C, H, W = x.shape
    if C != cin:
        raise RuntimeError(f'expected {cin} input channels, got {C}')
    ho, wo = ((H + 2 * p - k) // s + 1) * up, ((W + 2 * p - k) // s + 1) * up
    if out is None:
        out = torch.empty((B, cout, ho, wo), dtype=torch.float32, device=x.device)
        channel_offset = 0
    elif tuple(out.shape[2:]) != (ho, wo) or out.shape[0] != B:
        raise RuntimeError(f'layer output {(B, cout, ho, wo)} does not fit buffer {tuple(out.shape)}')
    fold = bn is None or (not bn.training and bn.running_mean is not None)
    if fold and bn is not None:
        scale, shift = cache.fold(conv, bn)
    elif conv.bias is not None:
        scale, shift = None, conv.bias.detach().float().contiguous()
    else:
        scale, shift = None, None
    fuse_relu = bool(relu) and fold
    L = _lib.lib()
    _lib.check(L.sf_conv2d_f32(_lib.ptr(x.contiguous()), cache.get(conv).data_ptr(),
                               _lib.ptr(scale), _lib.ptr(shift), out.data_ptr(), B, cin, H, W, cout,
                               k, s, p, up, out.shape[1], channel_offset, int(fuse_relu),
                               _lib.stream(x.device)))
    if not fold:   # training-mode BatchNorm: batch statistics through torch, on the device
        view = out[:, channel_offset:channel_offset + cout]
        y = bn(view)
        if relu:
            y = torch.relu_(y)
        view.copy_(y)
    return _lib.forward_only(out, x, conv.weight, conv.bias,
                             *((bn.weight, bn.bias) if bn is not None else ()))


def _run_sequential(seq, x, caches, out=None, channel_offset=0):
    """Walk an nn.Sequential of (conv, BatchNorm2d, ReLU) triples."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        conv = mods[i]
        if not isinstance(conv, (nn.Conv2d, nn.ConvTranspose2d)):
            raise NotImplementedError(f'unexpected layer {type(conv).__name__} in a SECOND block')
        bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d) else None
        j = i + (2 if bn is not None else 1)
        relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
        j += int(relu)
        last = j >= len(mods)
        x = conv_bn_relu(x, conv, bn, relu, caches.setdefault(id(conv), _PackedFilters()),
                         out if last else None, channel_offset if last else 0)
        i = j
    return x


@BACKBONES.register_module()
class SECOND(nn.Module):
    """second.py:8-86: per stage a strided 3x3 conv + `layer_num` 3x3 convs, each conv -> BN ->
    ReLU; returns the tuple of stage outputs."""

    def __init__(self, in_channels=128, out_channels=[128, 128, 256], layer_nums=[3, 5, 5],
                 layer_strides=[2, 2, 2], norm_cfg=dict(type='BN', eps=1e-3, momentum=0.01),
                 conv_cfg=dict(type='Conv2d', bias=False)):
        super().__init__()
        assert len(layer_strides) == len(layer_nums)
        assert len(out_channels) == len(layer_nums)
        in_filters = [in_channels, *out_channels[:-1]]
        blocks = []
        for i, layer_num in enumerate(layer_nums):
            block = [_build_conv(conv_cfg, in_filters[i], out_channels[i], 3,
                                 stride=layer_strides[i], padding=1),
                     _build_norm(norm_cfg, out_channels[i]), nn.ReLU(inplace=True)]
            for _ in range(layer_num):
                block.append(_build_conv(conv_cfg, out_channels[i], out_channels[i], 3, padding=1))
                block.append(_build_norm(norm_cfg, out_channels[i]))
                block.append(nn.ReLU(inplace=True))
            blocks.append(nn.Sequential(*block))
        self.blocks = nn.ModuleList(blocks)
        self._caches = {}

    def init_weights(self, pretrained=None):
        """second.py:64-71: conv layers keep torch's default init; checkpoints load through
        load_state_dict (mmcv's load_checkpoint is not available here)."""
        if isinstance(pretrained, str):
            sd = torch.load(pretrained, map_location='cpu', weights_only=True)
            self.load_state_dict(sd.get('state_dict', sd), strict=False)

    def forward(self, x):
        outs = []
        for block in self.blocks:
            x = _run_sequential(block, x, self._caches)
            outs.append(x)
        return tuple(outs)


@NECKS.register_module()
class SECONDFPN(nn.Module):
    """second_fpn.py:11-92: one deblock (deconv k = s = stride, or conv when the stride is <= 1 and
    `use_conv_for_no_stride`) -> BN -> ReLU per input scale, concatenated along channels."""

    def __init__(self, in_channels=[128, 128, 256], out_channels=[256, 256, 256],
                 upsample_strides=[1, 2, 4], norm_cfg=dict(type='BN', eps=1e-3, momentum=0.01),
                 upsample_cfg=dict(type='deconv', bias=False),
                 conv_cfg=dict(type='Conv2d', bias=False), use_conv_for_no_stride=False):
        super().__init__()
        assert len(out_channels) == len(upsample_strides) == len(in_channels)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.fp16_enabled = False
        deblocks = []
        for i, out_channel in enumerate(out_channels):
            stride = upsample_strides[i]
            if stride > 1 or (stride == 1 and not use_conv_for_no_stride):
                layer = _build_upsample(upsample_cfg, in_channels[i], out_channel,
                                        upsample_strides[i], upsample_strides[i])
            else:
                stride = int(np.round(1 / stride).astype(np.int64))
                layer = _build_conv(conv_cfg, in_channels[i], out_channel, stride, stride=stride)
            deblocks.append(nn.Sequential(layer, _build_norm(norm_cfg, out_channel),
                                          nn.ReLU(inplace=True)))
        self.deblocks = nn.ModuleList(deblocks)
        self._caches = {}

    def init_weights(self):
        """second_fpn.py:67-73."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _out_hw(self, i, x):
        layer = self.deblocks[i][0]
        k, s, p, up = _check_conv(layer)
        return ((x.shape[2] + 2 * p - k) // s + 1) * up, ((x.shape[3] + 2 * p - k) // s + 1) * up

    def forward(self, x, extra_channels=0):
        """second_fpn.py:76-92.  `extra_channels` (not in the reference) reserves that many
        uninitialised channels behind the concatenated ones, for a caller that appends maps of its
        own (SparseFusionDetector.extract_pts_feat, detectors/sparsefusion.py:127-130)."""
        assert len(x) == len(self.in_channels)
        sizes = {self._out_hw(i, xi) for i, xi in enumerate(x)}
        if len(sizes) != 1:
            raise RuntimeError(f'deblock outputs differ in size: {sorted(sizes)} (torch.cat would fail)')
        (ho, wo), = sizes
        out = torch.empty((x[0].shape[0], sum(self.out_channels) + int(extra_channels), ho, wo),
                          dtype=torch.float32,
                          device=x[0].device)
        off = 0
        for i, deblock in enumerate(self.deblocks):
            _run_sequential(deblock, x[i], self._caches, out, off)
            off += self.out_channels[i]
        return [out]

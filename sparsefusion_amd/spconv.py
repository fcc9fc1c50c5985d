"""Sparse-convolution operator API on MI355X: the surface the reference re-exports from
spconv (mmdet3d/ops/spconv/__init__.py:21-46) -- SparseConvTensor, SparseModule,
SparseSequential, SubMConv3d, SparseConv3d -- plus the functional layer of the vendored
spconv 1.x (mmdet3d/ops/spconv/ops.py:46-126): get_indice_pairs, indice_conv.

Everything numeric runs in libsparsefusion_hip.so.  The rulebook is kept in output-major form
(neighbour table) on the device and only converted to the reference's (K,2,N) pair tensors
when somebody asks for them.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from . import _lib


def _triple(v, name='value'):
    if isinstance(v, (list, tuple)):
        if len(v) != 3:
            raise ValueError(f'{name} must have 3 entries, got {v}')
        return [int(x) for x in v]
    return [int(v)] * 3


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    """ops/spconv/ops.py:20-30."""
    out = []
    for i in range(len(input_size)):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) \
            // stride[i] + 1
        out.append(1 if kernel_size[i] == -1 else size)
    return out


# ------------------------------------------------------------------ device-side structures

class CoordSet:
    """One active set: indices (N,4) int32 (b,z,y,x) on the device, its coordinate hash and the
    neighbour tables already derived from it (so SubM layers that share geometry share work;
    the reference rebuilds an identical rulebook for every un-keyed SubM conv)."""

    def __init__(self, indices, spatial_shape, batch_size, hash_table=None):
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self._hash = hash_table
        self.rulebooks = {}

    @property
    def n(self):
        return self.indices.shape[0]

    def hash(self):
        if self._hash is None:
            L = _lib.lib()
            dev = self.indices.device
            cap = L.sf_hash_capacity(max(self.n, 1))
            keys = torch.empty(cap, dtype=torch.int32, device=dev)
            vals = torch.empty(cap, dtype=torch.int32, device=dev)
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(L.sf_hash_build(_lib.ptr(self.indices), None, self.n,
                                       _lib.i3(self.spatial_shape), self.batch_size,
                                       keys.data_ptr(), vals.data_ptr(), cap, flag.data_ptr(),
                                       _lib.stream(dev)))
            self._hash = (keys, vals, cap)
        return self._hash


class Rulebook:
    """Neighbour table nbr (K, stride) int32 + what it connects."""

    def __init__(self, nbr, K, in_coords, out_coords, ksize, stride, padding, subm):
        self.nbr = nbr
        self.K = K
        self.in_coords = in_coords
        self.out_coords = out_coords
        self.ksize, self.stride, self.padding, self.subm = ksize, stride, padding, subm
        self._pairs = None

    @property
    def k_first(self):
        return (self.K - 1) // 2 if self.subm else -1

    def row_order(self):
        """Output rows sorted by neighbour mask (sf_row_order_by_mask), built once per rulebook:
        the tile schedule of sf_sparse_conv_f32_ordered."""
        if getattr(self, '_order', None) is None:
            L = _lib.lib()
            n = max(self.out_coords.n, 1)
            dev = self.nbr.device
            order = torch.empty(L.sf_row_order_size(n), dtype=torch.int32, device=dev)
            wsb = L.sf_row_order_workspace(n)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            _lib.check(L.sf_row_order_by_mask(self.nbr.data_ptr(), self.nbr.shape[1], self.K, self.k_first, None,
                                              self.out_coords.n, order.data_ptr(), ws.data_ptr(), wsb,
                                              _lib.stream(dev)))
            self._order = order
        return self._order

    def inverted(self):
        """Rulebook of the coupled inverse convolution (SparseInverseConv3d, conv.py:147-152 /
        spconv_ops.h:319-348 with inverse = 1): every pair (in, out) of offset k becomes (out, in)
        of the same offset, the output set is this rulebook's INPUT set.  A strided convolution
        reaches an input through one offset from at most one output, so the swapped pairs form a
        neighbour table again (built from the exported pairs; the kernels are the forward ones)."""
        if getattr(self, '_inverse', None) is None:
            if self.subm:
                raise RuntimeError('an inverse convolution couples with a SparseConv3d, not a SubM one')
            L = _lib.lib()
            pairs, num = self.indice_pairs()
            swapped = pairs[:, [1, 0], :].contiguous()
            n_new_out = self.in_coords.n
            stride = nbr_stride(n_new_out)
            nbr = torch.empty((self.K, stride), dtype=torch.int32, device=pairs.device)
            _lib.check(L.sf_indice_pairs_to_table(swapped.data_ptr(), num.data_ptr(), self.K,
                                                  swapped.shape[2], n_new_out, nbr.data_ptr(), stride,
                                                  _lib.stream(pairs.device)))
            inv = Rulebook(nbr, self.K, self.out_coords, self.in_coords, self.ksize, self.stride,
                           self.padding, False)
            inv._keep = swapped
            self._inverse = inv
        return self._inverse

    def indice_pairs(self):
        """(indice_pairs (K,2,N_in) int32, indice_num (K,) int32) in the reference layout."""
        if self._pairs is None:
            L = _lib.lib()
            n_in, n_out = self.in_coords.n, self.out_coords.n
            dev = self.nbr.device
            pairs = torch.empty((self.K, 2, n_in), dtype=torch.int32, device=dev)
            num = torch.empty((self.K,), dtype=torch.int32, device=dev)
            wsb = L.sf_export_indice_pairs_workspace(self.K, n_in)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            _lib.check(L.sf_export_indice_pairs(self.nbr.data_ptr(), self.nbr.shape[1], self.K,
                                                None, n_out, n_in, pairs.data_ptr(),
                                                num.data_ptr(), ws.data_ptr(), wsb,
                                                _lib.stream(dev)))
            self._pairs = (pairs, num)
        return self._pairs


# fp32 layer shapes (padded CIN x COUT) of SubM rulebooks that run on the mask-ordered kernel: the
# measured winners (tools/ab_layers.sh, bench.py).  'all' / 'none' / a comma list; a
# module attribute so that a test or an A/B tool can set it before a LidarBackbone is built.
ORDERED_SHAPES = '64x64,128x128'


def use_ordered(cin_padded, cout, subm=True):
    """Whether an fp32 layer of this shape runs on the mask-ordered kernel
    (sf_sparse_conv_f32_ordered) instead of the pair-compacting / direct ones (ORDERED_SHAPES).
    Unless ORDERED_SHAPES is 'all', only SubM layers qualify: their rulebook (and so the plan the
    kernel needs) is shared by the four convolutions of a stage, a strided rulebook is used once
    and its sort would cost more than it saves."""
    sel = ORDERED_SHAPES.strip().lower()
    if sel != 'all' and not subm:
        return False
    if cin_padded < 16 or sel in ('', 'none'):
        return False
    if sel == 'all':
        return True
    return f'{cin_padded}x{cout}' in sel.split(',')


def nbr_stride(rows):
    """Column stride (in rows) of a neighbour table for `rows` output rows: a multiple of 1024
    rows plus 64, so that the K columns a tile reads together start 256 bytes apart modulo 4 KiB
    and spread over the memory channels (a stride that is a multiple of 4 KiB puts all 27 on
    one: measured 2.4x slower narrow layers at batch 8)."""
    return (max(int(rows), 1) + 1023) // 1024 * 1024 + 64


def build_rulebook(coords, ksize, stride, padding, dilation, subm):
    """Rulebook of one conv over `coords` (ops.get_indice_pairs semantics, spconv_ops.h:27-141)."""
    L = _lib.lib()
    ks, st, pd, dl = _triple(ksize), _triple(stride), _triple(padding), _triple(dilation)
    if any(d != 1 for d in dl):
        raise NotImplementedError('dilation != 1 is not supported (SparseEncoder never uses it)')
    if subm:
        st, pd = [1, 1, 1], [k // 2 for k in ks]  # spconv_ops.h:76-79
    key = (tuple(ks), tuple(st), tuple(pd), bool(subm))
    if key in coords.rulebooks:
        return coords.rulebooks[key]
    K = ks[0] * ks[1] * ks[2]
    if K > 27:
        raise NotImplementedError('kernel volume > 27 is not supported')
    dev = coords.indices.device
    stream = _lib.stream(dev)
    keys, vals, cap = coords.hash()
    if subm:
        out_coords = coords
    else:
        out_shape = get_conv_output_size(coords.spatial_shape, ks, st, pd, dl)
        vol = coords.batch_size * out_shape[0] * out_shape[1] * out_shape[2]
        reach = 1
        for k, s in zip(ks, st):
            reach *= (k + s - 1) // s  # outputs one input can touch per dim
        bound = max(min(coords.n * reach, vol), 1)
        ocap = L.sf_hash_capacity(bound)
        okeys = torch.empty(ocap, dtype=torch.int32, device=dev)
        ovals = torch.empty(ocap, dtype=torch.int32, device=dev)
        outids = torch.empty((bound, 4), dtype=torch.int32, device=dev)
        cnt = torch.zeros(2, dtype=torch.int32, device=dev)  # [n_out, overflow]
        wsb = L.sf_strided_rulebook_workspace(coords.n, ocap, _lib.i3(ks), _lib.i3(st))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        # output set + neighbour table in one call, sized for the bound; trimmed below once the
        # row count is known (the one host sync of a strided conv in module mode)
        wide = torch.empty((K, nbr_stride(bound)), dtype=torch.int32, device=dev)
        _lib.check(L.sf_strided_rulebook(_lib.ptr(coords.indices), None, coords.n,
                                         coords.batch_size, _lib.i3(out_shape), _lib.i3(ks),
                                         _lib.i3(st), _lib.i3(pd), okeys.data_ptr(),
                                         ovals.data_ptr(), ocap, outids.data_ptr(), bound,
                                         cnt.data_ptr(), cnt.data_ptr() + 4, wide.data_ptr(),
                                         wide.shape[1], ws.data_ptr(), wsb, stream))
        n_out, overflow = cnt.tolist()
        if overflow:
            raise RuntimeError('sparse conv output set overflowed its table')
        out_coords = CoordSet(outids[:n_out], out_shape, coords.batch_size, (okeys, ovals, ocap))
        nbr = torch.empty((K, nbr_stride(n_out)), dtype=torch.int32, device=dev)
        nbr[:, :n_out] = wide[:, :n_out]
        del wide
        rb = Rulebook(nbr, K, coords, out_coords, ks, st, pd, False)
        coords.rulebooks[key] = rb
        return rb
    n_out = out_coords.n
    stride_rows = nbr_stride(n_out)
    nbr = torch.empty((K, stride_rows), dtype=torch.int32, device=dev)
    _lib.check(L.sf_neighbor_table(_lib.ptr(out_coords.indices), None, n_out, coords.batch_size,
                                   _lib.i3(coords.spatial_shape), _lib.i3(ks), _lib.i3(st),
                                   _lib.i3(pd), keys.data_ptr(), vals.data_ptr(), cap,
                                   1 if subm else 0, nbr.data_ptr(), stride_rows, stream))
    rb = Rulebook(nbr, K, coords, out_coords, ks, st, pd, bool(subm))
    coords.rulebooks[key] = rb
    return rb


class IndiceData(tuple):
    """indice_dict entry.  Unpacks like the reference's 5-tuple
    (outids, indices, indice_pairs, indice_pair_num, spatial_shape) (conv.py:169-172); the pair
    tensors are exported from the neighbour table on first access."""

    def __new__(cls, rulebook):
        self = super().__new__(cls, ())
        self.rulebook = rulebook
        return self

    def _fields(self):
        rb = self.rulebook
        pairs, num = rb.indice_pairs()
        return (rb.out_coords.indices, rb.in_coords.indices, pairs, num,
                rb.in_coords.spatial_shape)

    def __iter__(self):
        return iter(self._fields())

    def __len__(self):
        return 5

    def __getitem__(self, i):
        return self._fields()[i]


class SparseConvTensor:
    """structure.py:21-69 / spconv.pytorch.core.SparseConvTensor (2.x adds replace_feature)."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        if indices.dtype != torch.int32:
            indices = indices.int()
        self.indices = indices.contiguous()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)  # accepts a 0-d tensor (detectors/sparsefusion.py:121)
        self.indice_dict = {}
        self.grid = grid
        self._coords = None

    def coords(self):
        if self._coords is None:
            self._coords = CoordSet(self.indices, self.spatial_shape, self.batch_size)
        return self._coords

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def replace_feature(self, features):
        out = SparseConvTensor(features, self.indices, self.spatial_shape, self.batch_size,
                               self.grid)
        out.indice_dict = self.indice_dict
        out._coords = self._coords
        return out

    def dense(self, channels_first=True):
        """(B, C, D, H, W) (or (B, D, H, W, C)); structure.py:55-64."""
        feats = self.features
        _lib.require(feats, torch.float32, 'features')
        C = feats.shape[1]
        D, H, W = self.spatial_shape
        out = torch.empty((self.batch_size, C, D, H, W), dtype=feats.dtype, device=feats.device)
        _lib.check(_lib.lib().sf_dense_bev(_lib.ptr(feats.contiguous()), _lib.ptr(self.indices),
                                           None, feats.shape[0], C, self.batch_size, D, H, W,
                                           out.data_ptr(), _lib.stream(feats.device)))
        if channels_first:
            return out
        return out.permute(0, 2, 3, 4, 1).contiguous()

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size


# ------------------------------------------------------------------ functional layer

def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                     out_padding=0, subm=False, transpose=False, grid=None):
    """ops/spconv/ops.py:46-105 -> (outids, indice_pairs (K,2,N), indice_pair_num (K,)).
    Output rows in the CPU reference's first-touch order, pairs in input-row order."""
    if transpose:
        raise NotImplementedError('transposed (inverse) convolution is outside the hot path')
    coords = CoordSet(indices.int().contiguous(), spatial_shape, batch_size)
    rb = build_rulebook(coords, ksize, stride, padding, dilation, subm)
    pairs, num = rb.indice_pairs()
    return rb.out_coords.indices, pairs, num


def pack_filters(weight_kio):
    """(K, Cin, Cout) fp32 on the device -> MFMA fragment order for sf_sparse_conv_f32."""
    L = _lib.lib()
    K, cin, cout = weight_kio.shape
    cp = L.sf_conv_padded_cin(cin)
    if cp < 0:
        raise RuntimeError(f'unsupported in_channels {cin}')
    packed = torch.empty(K * max(cp, 16) * cout, dtype=torch.float32, device=weight_kio.device)
    _lib.check(L.sf_pack_filters_f32(_lib.ptr(weight_kio.contiguous()), K, cin, cout,
                                     packed.data_ptr(), _lib.stream(weight_kio.device)))
    return packed


def pack_filters_bf16(weight_kio):
    """(K, Cin, Cout) fp32 on the device -> bf16 MFMA fragment order for sf_sparse_conv_bf16."""
    L = _lib.lib()
    K, cin, cout = weight_kio.shape
    cp = L.sf_conv_padded_cin_bf16(cin)
    if cp < 0:
        raise RuntimeError(f'unsupported in_channels {cin}')
    packed = torch.empty(K * cp * cout, dtype=torch.bfloat16, device=weight_kio.device)
    _lib.check(L.sf_pack_filters_bf16(_lib.ptr(weight_kio.contiguous()), K, cin, cout,
                                      packed.data_ptr(), _lib.stream(weight_kio.device)))
    return packed


def cast_bf16(features, channels=None):
    """(N, C) fp32 -> (N, channels >= C) bf16, round to nearest even, extra columns zero: the
    explicit form of auto_fp16's cast of voxel_features (sparse_encoder.py:95)."""
    _lib.require(features, torch.float32, 'features')
    n, c = features.shape
    cd = c if channels is None else int(channels)
    out = torch.empty((n, cd), dtype=torch.bfloat16, device=features.device)
    _lib.check(_lib.lib().sf_cast_f32_bf16(_lib.ptr(features.contiguous()), c, out.data_ptr(), cd,
                                           None, n, _lib.stream(features.device)))
    return out


def _padded_features_bf16(features, cin):
    cp = _lib.lib().sf_conv_padded_cin_bf16(cin)
    if cp < 0:
        raise RuntimeError(f'unsupported in_channels {cin}')
    if features.shape[1] == cp:
        return features.contiguous(), cp
    dst = torch.zeros((features.shape[0], cp), dtype=torch.bfloat16, device=features.device)
    dst[:, :features.shape[1]] = features
    return dst, cp


def _padded_features(features, cin):
    L = _lib.lib()
    cp = L.sf_conv_padded_cin(cin)
    if cp < 0:
        raise RuntimeError(f'unsupported in_channels {cin}')
    if features.shape[1] == cp:
        return features.contiguous(), cp
    dst = torch.empty((features.shape[0], cp), dtype=torch.float32, device=features.device)
    _lib.check(L.sf_pad_channels_f32(_lib.ptr(features.contiguous()), features.shape[1],
                                     dst.data_ptr(), cp, None, features.shape[0],
                                     _lib.stream(features.device)))
    return dst, cp


def sparse_conv(features, packed_filters, cin, cout, rulebook, scale=None, shift=None,
                residual=None, relu=False, out_dtype=None, params=()):
    """One fused layer: conv (+ folded BN scale/shift) (+ residual) (+ ReLU).  fp32 features run
    the exact-fp32 kernel; bf16 features (with bf16-packed filters) the bf16 MFMA kernel, whose
    output is bf16 unless out_dtype=torch.float32."""
    L = _lib.lib()
    if features.shape[0] != rulebook.in_coords.n:
        raise RuntimeError('features rows do not match the rulebook input set')
    # `params`: the nn.Parameters the packed filters / folded scale and shift were derived from
    grad_deps = (features, residual) + tuple(params)
    if features.dtype == torch.bfloat16:
        _lib.require(packed_filters, torch.bfloat16, 'packed filters')
        x, cp = _padded_features_bf16(features, cin)
        n_out = rulebook.out_coords.n
        odt = torch.bfloat16 if out_dtype is None else out_dtype
        if odt not in (torch.bfloat16, torch.float32):
            raise RuntimeError(f'unsupported out_dtype {odt}')
        out = torch.empty((n_out, cout), dtype=odt, device=features.device)
        if residual is not None:
            if residual.shape != out.shape:
                raise RuntimeError('residual shape mismatch')
            residual = _lib.require(residual, torch.bfloat16, 'residual').contiguous()
        _lib.check(L.sf_sparse_conv_bf16(
            _lib.ptr(x), x.shape[0], cp, rulebook.nbr.data_ptr(), rulebook.nbr.shape[1], rulebook.K,
            rulebook.k_first, _lib.ptr(packed_filters), cout, _lib.ptr(scale), _lib.ptr(shift),
            _lib.ptr(residual), 1 if relu else 0, out.data_ptr(), 1 if odt == torch.float32 else 0,
            None, n_out, _lib.stream(features.device)))
        return _lib.forward_only(out, *grad_deps)
    _lib.require(features, torch.float32, 'features')
    if out_dtype not in (None, torch.float32):
        raise RuntimeError('fp32 features give fp32 output (cast with cast_bf16)')
    x, cp = _padded_features(features, cin)
    n_out = rulebook.out_coords.n
    out = torch.empty((n_out, cout), dtype=torch.float32, device=features.device)
    if residual is not None:
        if residual.shape != out.shape:
            raise RuntimeError('residual shape mismatch')
        residual = residual.contiguous()
    if use_ordered(cp, cout, rulebook.subm) and n_out > 0:   # same selection as LidarBackbone: identical results
        _lib.check(L.sf_sparse_conv_f32_ordered(
            _lib.ptr(x), x.shape[0], cp, rulebook.nbr.data_ptr(), rulebook.nbr.shape[1], rulebook.K,
            rulebook.k_first, rulebook.row_order().data_ptr(), _lib.ptr(packed_filters), cout,
            _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(residual), 1 if relu else 0, out.data_ptr(),
            None, n_out, _lib.stream(features.device)))
        return _lib.forward_only(out, *grad_deps)
    _lib.check(L.sf_sparse_conv_f32(
        _lib.ptr(x), x.shape[0], cp, rulebook.nbr.data_ptr(), rulebook.nbr.shape[1], rulebook.K,
        rulebook.k_first, _lib.ptr(packed_filters), cout, _lib.ptr(scale), _lib.ptr(shift),
        _lib.ptr(residual), 1 if relu else 0, out.data_ptr(), None, n_out,
        _lib.stream(features.device)))
    return _lib.forward_only(out, *grad_deps)


def indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse=False,
                subm=False):
    """ops/spconv/ops.py:108-126 (indice_conv_fp32): filters (kD,kH,kW,Cin,Cout), rulebook in
    the reference's pair format (e.g. straight from get_indice_pairs)."""
    if inverse:
        raise NotImplementedError('inverse convolution is outside the hot path')
    L = _lib.lib()
    cin, cout = filters.shape[-2], filters.shape[-1]
    K, _, Lp = indice_pairs.shape
    dev = features.device
    n_out = int(num_activate_out)
    stride_rows = max(n_out, 1)
    nbr = torch.empty((K, stride_rows), dtype=torch.int32, device=dev)
    _lib.check(L.sf_indice_pairs_to_table(_lib.ptr(indice_pairs.int().contiguous()),
                                          _lib.ptr(indice_pair_num.int().contiguous()), K, Lp,
                                          n_out, nbr.data_ptr(), stride_rows, _lib.stream(dev)))
    packed = pack_filters(filters.reshape(K, cin, cout).float())
    x, cp = _padded_features(features, cin)
    out = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
    k_first = (K - 1) // 2 if subm else -1
    _lib.check(L.sf_sparse_conv_f32(_lib.ptr(x), x.shape[0], cp, nbr.data_ptr(), stride_rows, K, k_first,
                                    packed.data_ptr(), cout, None, None, None, 0, out.data_ptr(),
                                    None, n_out, _lib.stream(dev)))
    return _lib.forward_only(out, features, filters)


# ------------------------------------------------------------------ modules

class SparseModule(nn.Module):
    """Marker base class: modules that take a SparseConvTensor (modules.py:25-28)."""


def fold_bn(bn):
    """Eval-mode BatchNorm1d -> per-channel (scale, shift): scale = gamma/sqrt(var+eps),
    shift = beta - mean*scale (the correct fold, tools/misc/fuse_conv_bn.py:18-21)."""
    var, mean = bn.running_var.float(), bn.running_mean.float()
    scale = torch.rsqrt(var + bn.eps)
    if bn.weight is not None:
        scale = scale * bn.weight.float()
    shift = -mean * scale
    if bn.bias is not None:
        shift = shift + bn.bias.float()
    return scale.contiguous(), shift.contiguous()


def _bn_foldable(m):
    return (isinstance(m, nn.BatchNorm1d) and not m.training and m.running_mean is not None and
            m.running_var is not None)


class SparseConvolution(SparseModule):
    """conv.py:48-204.  Parameter layout is spconv-2's (Cout, kD, kH, kW, Cin), the layout the
    reference's checkpoints hold (write_spconv2.py:46-47); a spconv-1 tensor
    (kD,kH,kW,Cin,Cout) in a state dict is permuted on load."""

    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0,
                 dilation=1, groups=1, bias=True, subm=False, output_padding=0, transposed=False,
                 inverse=False, indice_key=None, fused_bn=False):
        super().__init__()
        if ndim != 3:
            raise NotImplementedError('only 3-D sparse convolution is on the hot path')
        if groups != 1 or transposed:
            raise NotImplementedError('groups/transposed convolution is out of scope')
        self.ndim = ndim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _triple(kernel_size, 'kernel_size')
        self.conv1x1 = int(np.prod(self.kernel_size)) == 1
        self.stride = _triple(stride, 'stride')
        self.padding = _triple(padding, 'padding')
        self.dilation = _triple(dilation, 'dilation')
        self.output_padding = _triple(output_padding, 'output_padding')
        self.transposed, self.inverse, self.groups = transposed, inverse, groups
        self.subm = subm
        self.indice_key = indice_key
        self.fused_bn = fused_bn
        self.weight = nn.Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self._packed = {}
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * int(np.prod(self.kernel_size))
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        w = state_dict.get(prefix + 'weight')
        v2 = (self.out_channels, *self.kernel_size, self.in_channels)
        v1 = (*self.kernel_size, self.in_channels, self.out_channels)
        if w is not None and tuple(w.shape) != v2 and tuple(w.shape) == v1:
            state_dict[prefix + 'weight'] = w.permute(4, 0, 1, 2, 3).contiguous()
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def packed_filters(self, dtype=torch.float32):
        key = (self.weight._version, self.weight.device, self.weight.data_ptr())
        hit = self._packed.get(dtype)
        if hit is None or hit[0] != key:
            K = int(np.prod(self.kernel_size))
            w = self.weight.detach().float().permute(1, 2, 3, 4, 0).reshape(
                K, self.in_channels, self.out_channels).contiguous()
            hit = (key, pack_filters_bf16(w) if dtype == torch.bfloat16 else pack_filters(w))
            self._packed[dtype] = hit
        return hit[1]

    def rulebook_for(self, input):
        datas = input.find_indice_pair(self.indice_key)
        if self.inverse:   # conv.py:147-152: the couple conv's rulebook, pairs swapped
            assert datas is not None and self.indice_key is not None
            rb = datas.rulebook
            assert rb.K == int(np.prod(self.kernel_size)), \
                'inverse conv must have same kernel size as its couple conv'
            return rb.inverted()
        if self.indice_key is not None and datas is not None:
            return datas.rulebook
        rb = build_rulebook(input.coords(), self.kernel_size, self.stride, self.padding,
                            self.dilation, self.subm)
        if self.indice_key is not None:
            input.indice_dict[self.indice_key] = IndiceData(rb)
        return rb

    def forward_fused(self, input, scale=None, shift=None, residual=None, relu=False,
                      out_dtype=None, bn_params=()):
        """bn_params: the BatchNorm affine parameters folded into scale / shift, so that autograd
        sees them (the op is forward-only: its backward raises)."""
        assert isinstance(input, SparseConvTensor)
        rb = self.rulebook_for(input)
        if self.bias is not None:
            b = self.bias.detach().float()
            shift = b if shift is None else shift + b * scale
            if scale is None:
                scale = torch.ones_like(b)
        feats = sparse_conv(input.features, self.packed_filters(input.features.dtype),
                            self.in_channels, self.out_channels, rb, scale, shift, residual, relu,
                            out_dtype, params=(self.weight, self.bias) + tuple(bn_params))
        out = SparseConvTensor(feats, rb.out_coords.indices, rb.out_coords.spatial_shape,
                               input.batch_size, input.grid)
        out.indice_dict = input.indice_dict
        out._coords = rb.out_coords
        return out

    def forward(self, input):
        return self.forward_fused(input)


class SparseConv3d(SparseConvolution):
    """conv.py:233-256."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    """conv.py:359-374: the inverse of the SparseConv3d that registered `indice_key`; restores that
    convolution's input index set (rows in the same order)."""

    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True,
                         indice_key=indice_key)


class SubMConv3d(SparseConvolution):
    """conv.py:404-428."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, True, indice_key=indice_key)


class SparseSequential(SparseModule):
    """modules.py:55-137: nn.Sequential that threads a SparseConvTensor through SparseModules and
    applies plain nn.Modules to `.features`.  On top of that contract, a
    conv -> BatchNorm1d(eval) [-> ReLU] run is executed as ONE fused kernel."""

    def __init__(self, *layers, **named_layers):
        """Accepts what the reference container accepts (modules.py:55-123): positional modules
        (registered as '0', '1', ...), ONE ordered mapping name -> module, and/or keyword
        modules; a keyword that repeats an existing name is an error."""
        super().__init__()
        if len(layers) == 1 and isinstance(layers[0], OrderedDict):
            entries = list(layers[0].items())
        else:
            entries = [(str(pos), layer) for pos, layer in enumerate(layers)]
        for name, layer in entries + list(named_layers.items()):
            self._register_unique(name, layer)

    def _register_unique(self, name, layer):
        if name in self._modules:
            raise ValueError(f'SparseSequential already holds a layer called {name!r}')
        self.add_module(name, layer)

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, position):
        """Layer at `position` in registration order; negative positions count from the end."""
        count = len(self)
        if isinstance(position, slice):
            raise TypeError('SparseSequential supports integer positions only')
        resolved = position + count if position < 0 else position
        if not 0 <= resolved < count:
            raise IndexError(f'position {position} outside a SparseSequential of {count} layers')
        for i, layer in enumerate(self._modules.values()):
            if i == resolved:
                return layer

    def add(self, module, name=None):
        """Append a layer; without a name it takes the next position number."""
        self._register_unique(str(len(self)) if name is None else name, module)

    def forward(self, input, out_dtype=None):
        """out_dtype: dtype of the features a fused conv -> BN [-> ReLU] run that ENDS the sequence
        writes (bf16 inputs only; SparseEncoder asks its last layer for fp32)."""
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, SparseConvolution) and isinstance(input, SparseConvTensor) and \
                    i + 1 < len(mods) and _bn_foldable(mods[i + 1]):
                scale, shift = fold_bn(mods[i + 1])
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                last = i + (3 if relu else 2) >= len(mods)
                input = m.forward_fused(input, scale, shift, None, relu,
                                        out_dtype if last else None,
                                        bn_params=(mods[i + 1].weight, mods[i + 1].bias))
                i += 3 if relu else 2
            elif isinstance(m, SparseModule):
                assert isinstance(input, SparseConvTensor)
                input = m(input)
                i += 1
            else:
                if isinstance(input, SparseConvTensor):
                    if input.indices.shape[0] != 0:
                        input = input.replace_feature(m(input.features))
                else:
                    input = m(input)
                i += 1
        return input


from .registry import CONV_LAYERS  # noqa: E402

CONV_LAYERS.register_module('SparseConv3d', module=SparseConv3d, force=True)
CONV_LAYERS.register_module('SubMConv3d', module=SubMConv3d, force=True)
CONV_LAYERS.register_module('SparseInverseConv3d', module=SparseInverseConv3d, force=True)

"""numpy/ctypes front-end of the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  It wraps oracle/liboracle.so (plain-C restatement of the reference's CPU
algorithms, see sf_oracle.c) and restates the layer wiring of the reference's
SparseEncoder (mmdet3d/models/middle_encoders/sparse_encoder.py:31-203,
mmdet3d/ops/sparse_block.py:116-204, mmdet3d/ops/spconv/conv.py:113-204).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags='C_CONTIGUOUS')
_i32p = np.ctypeslib.ndpointer(np.int32, flags='C_CONTIGUOUS')


def build():
    """Compile liboracle.so (gcc).  Building the checker is not using it."""
    subprocess.check_call(['make', '-s', '-C', _HERE, 'liboracle.so'])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.sfo_grid_size.argtypes = [_f32p, _f32p, _i32p]
        L.sfo_dynamic_voxelize.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p, _i32p]
        L.sfo_hard_voxelize.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int,
                                        _f32p, _i32p, _i32p]
        L.sfo_hard_voxelize.restype = C.c_int
        L.sfo_vfe_mean.argtypes = [_f32p, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        L.sfo_voxel_heights.argtypes = [_f32p, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, _f32p, _f32p]
        L.sfo_indice_pairs_subm.argtypes = [_i32p, C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p,
                                            _i32p]
        L.sfo_indice_pairs_subm.restype = C.c_int
        L.sfo_indice_pairs_conv.argtypes = [_i32p, C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p,
                                            _i32p, _i32p, _i32p, _i32p]
        L.sfo_indice_pairs_conv.restype = C.c_int
        L.sfo_indice_conv.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, _i32p, _i32p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        L.sfo_indice_conv.restype = C.c_int
        L.sfo_bn_act.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_float,
                                 C.c_void_p, C.c_int]
        L.sfo_dynamic_scatter.argtypes = [_f32p, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p,
                                          _i32p, _i32p, _i32p]
        L.sfo_dynamic_scatter.restype = C.c_int
        L.sfo_dense_bev.argtypes = [_f32p, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, _f32p]
        L.sfo_num_threads.restype = C.c_int
        L.sfo_set_num_threads.argtypes = [C.c_int]
        L.sfo_conv_stage_times.argtypes = [C.POINTER(C.c_double), C.c_int]
        _LIB = L
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3
        return [int(x) for x in v]
    return [int(v)] * 3


# ---------------------------------------------------------------- voxelize

def grid_size(voxel_size, coors_range):
    g = np.zeros(3, np.int32)
    lib().sfo_grid_size(_f32(voxel_size), _f32(coors_range), g)
    return g


def dynamic_voxelize(points, voxel_size, coors_range):
    """-> coors (N,3) int32 (z,y,x), -1 rows for rejected points."""
    points = _f32(points)
    coors = np.zeros((points.shape[0], 3), np.int32)
    lib().sfo_dynamic_voxelize(points, points.shape[0], points.shape[1], _f32(voxel_size),
                               _f32(coors_range), coors)
    return coors


def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    """-> (voxels (M,max_points,F), coors (M,3), num_points (M,)) as voxelize.py:46-58."""
    points = _f32(points)
    n, f = points.shape
    voxels = np.zeros((max_voxels, max_points, f), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    npv = np.zeros((max_voxels,), np.int32)
    m = lib().sfo_hard_voxelize(points, n, f, _f32(voxel_size), _f32(coors_range), max_points,
                                max_voxels, voxels, coors, npv)
    if m < 0:
        raise MemoryError('oracle dense grid allocation failed')
    return voxels[:m], coors[:m], npv[:m]


def vfe_mean(voxels, num_points, num_features):
    voxels = _f32(voxels)
    m, mp, f = voxels.shape
    out = np.zeros((m, num_features), np.float32)
    if m:
        lib().sfo_vfe_mean(voxels, _i32(num_points), m, mp, f, num_features, out)
    return out


def voxel_heights(voxels, coors4, batch, grid_xyz, out_size_factor):
    voxels = _f32(voxels)
    m, mp, f = voxels.shape
    nx, ny = grid_xyz[0] // out_size_factor, grid_xyz[1] // out_size_factor
    mn = np.zeros((batch, ny, nx), np.float32)
    mx = np.zeros((batch, ny, nx), np.float32)
    lib().sfo_voxel_heights(voxels, _i32(coors4), m, mp, f, batch, ny, nx, out_size_factor, mn, mx)
    return mn, mx


def vfe_layer(inputs, weight, bn, eps=1e-3, max_out=True, cat_max=True):
    """VFELayer.forward (mmdet3d/models/voxel_encoders/utils.py:61-96) with eval-mode BatchNorm1d:
    inputs (M, P, C), weight (Cout, C) = nn.Linear.weight, bn = dict(weight, bias, running_mean,
    running_var)."""
    x = _f32(inputs) @ _f32(weight).T                                      # :85  linear, no bias
    x = (x - _f32(bn['running_mean'])) / np.sqrt(_f32(bn['running_var']) + np.float32(eps)) \
        * _f32(bn['weight']) + _f32(bn['bias'])                            # :86  norm over channels
    pointwise = np.maximum(x, 0).astype(np.float32)                        # :88
    if not max_out:
        return pointwise                                                   # :93-95
    aggregated = pointwise.max(axis=1, keepdims=True)                      # :91  ALL P slots
    if not cat_max:
        return aggregated[:, 0]                                            # :97-98
    return np.concatenate([pointwise, np.repeat(aggregated, inputs.shape[1], 1)], 2)   # :100-104


def hard_vfe_decorate(features, num_points, coors, voxel_size, point_cloud_range,
                      with_cluster_center=False, with_voxel_center=False):
    """HardVFE.forward up to the VFE layers (mmdet3d/models/voxel_encoders/voxel_encoder.py:404-441):
    decorations, then zero every slot >= num_points.  coors rows are (b, z, y, x)."""
    f = _f32(features)
    ls = [f]
    if with_cluster_center:                                                # :407-413
        mean = f[:, :, :3].sum(axis=1, keepdims=True) / \
            np.asarray(num_points, np.float32).reshape(-1, 1, 1)
        ls.append(f[:, :, :3] - mean)
    if with_voxel_center:                                                  # :416-428
        vx, vy, vz = (np.float32(v) for v in voxel_size)
        off = [np.float32(voxel_size[i] / 2 + point_cloud_range[i]) for i in range(3)]
        c = np.asarray(coors)
        fc = np.zeros(f.shape[:2] + (3,), np.float32)
        fc[:, :, 0] = f[:, :, 0] - (c[:, 3].astype(np.float32)[:, None] * vx + off[0])
        fc[:, :, 1] = f[:, :, 1] - (c[:, 2].astype(np.float32)[:, None] * vy + off[1])
        fc[:, :, 2] = f[:, :, 2] - (c[:, 1].astype(np.float32)[:, None] * vz + off[2])
        ls.append(fc)
    x = np.concatenate(ls, -1)                                             # :435
    mask = np.asarray(num_points).reshape(-1, 1) > np.arange(f.shape[1]).reshape(1, -1)   # :439-441
    return (x * mask[:, :, None]).astype(np.float32)


def hard_vfe(features, num_points, coors, layers, voxel_size, point_cloud_range, eps=1e-3,
             with_cluster_center=False, with_voxel_center=False):
    """HardVFE.forward (:383-450) without a fusion layer.  layers = [(weight, bn), ...]; every
    layer but the last concatenates (cat_max), the last returns the per-voxel max (:349-361)."""
    x = hard_vfe_decorate(features, num_points, coors, voxel_size, point_cloud_range,
                          with_cluster_center, with_voxel_center)
    for i, (w, bn) in enumerate(layers):
        x = vfe_layer(x, w, bn, eps, True, i != len(layers) - 1)
    return x


def pillar_decorate(features, num_points, coors, voxel_size, point_cloud_range,
                    with_cluster_center=True, with_voxel_center=True, with_distance=False,
                    legacy=True):
    """PillarFeatureNet.forward up to the PFN layers (mmdet3d/models/voxel_encoders/
    pillar_encoder.py:103-145).  In legacy mode the reference's f_center is a VIEW of
    features[:, :, :2] that is modified in place (:124-130), so the raw x / y columns carry the
    offsets as well -- after f_cluster was computed (:105-110), before the distance is (:133-135)."""
    f = _f32(features).copy()
    ls = [f]
    if with_cluster_center:                                                # :105-110
        mean = f[:, :, :3].sum(axis=1, keepdims=True) / \
            np.asarray(num_points, np.float32).reshape(-1, 1, 1)
        ls.append(f[:, :, :3] - mean)
    if with_voxel_center:                                                  # :114-131
        vx, vy = np.float32(voxel_size[0]), np.float32(voxel_size[1])
        xo = np.float32(voxel_size[0] / 2 + point_cloud_range[0])          # :86-87
        yo = np.float32(voxel_size[1] / 2 + point_cloud_range[1])
        c = np.asarray(coors)
        fc = np.zeros(f.shape[:2] + (2,), np.float32)
        fc[:, :, 0] = f[:, :, 0] - (c[:, 3].astype(np.float32)[:, None] * vx + xo)
        fc[:, :, 1] = f[:, :, 1] - (c[:, 2].astype(np.float32)[:, None] * vy + yo)
        if legacy:
            f[:, :, :2] = fc                                               # the aliasing of :124
        ls.append(fc)
    if with_distance:                                                      # :133-135
        ls.append(np.sqrt((f[:, :, :3] ** 2).sum(-1, keepdims=True)).astype(np.float32))
    x = np.concatenate(ls, -1)                                             # :138
    mask = np.asarray(num_points).reshape(-1, 1) > np.arange(f.shape[1]).reshape(1, -1)   # :142-145
    return (x * mask[:, :, None]).astype(np.float32)


def pillar_feature_net(features, num_points, coors, layers, voxel_size, point_cloud_range, eps=1e-3,
                       with_cluster_center=True, with_voxel_center=True, with_distance=False,
                       legacy=True):
    """PillarFeatureNet.forward (:91-150), mode 'max'.  layers = [(weight, bn), ...]; a PFNLayer
    (voxel_encoders/utils.py:159-180) is Linear -> BN -> ReLU -> max, concatenated with the
    point-wise features unless it is the last one -- the arithmetic of vfe_layer."""
    x = pillar_decorate(features, num_points, coors, voxel_size, point_cloud_range,
                        with_cluster_center, with_voxel_center, with_distance, legacy)
    for i, (w, bn) in enumerate(layers):
        x = vfe_layer(x, w, bn, eps, True, i != len(layers) - 1)
    return x


def pillar_scatter(voxel_features, coors4, batch_size, ny, nx):
    """PointPillarsScatter.forward_batch (mmdet3d/models/middle_encoders/pillar_scatter.py:62-101):
    canvas[b][:, y * nx + x] = features, coors rows (b, z, y, x)."""
    c = np.asarray(coors4)
    out = np.zeros((batch_size, voxel_features.shape[1], ny * nx), np.float32)
    for b in range(batch_size):
        m = c[:, 0] == b
        out[b][:, c[m, 2] * nx + c[m, 3]] = _f32(voxel_features)[m].T
    return out.reshape(batch_size, voxel_features.shape[1], ny, nx)


def batch_coors(coors_list):
    """detectors/sparsefusion.py:153-157: prepend the sample index."""
    out = []
    for i, c in enumerate(coors_list):
        out.append(np.concatenate([np.full((c.shape[0], 1), i, np.int32), c.astype(np.int32)], 1))
    return np.concatenate(out, 0) if out else np.zeros((0, 4), np.int32)


# ---------------------------------------------------------------- rulebook

def conv_output_size(in_shape, ksize, stride, padding, dilation):
    """ops/spconv/ops.py:20-30."""
    return [(in_shape[i] + 2 * padding[i] - dilation[i] * (ksize[i] - 1) - 1) // stride[i] + 1
            for i in range(3)]


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                     subm=False):
    """ops/spconv/ops.py:46-105 -> (outids (M,4), indice_pairs (K,2,N), indice_num (K,), out_shape).

    CPU reference order: outputs in first-touch order, pairs in input order."""
    indices = _i32(indices)
    n = indices.shape[0]
    ks, st, pd, dl = _triple(ksize), _triple(stride), _triple(padding), _triple(dilation)
    K = ks[0] * ks[1] * ks[2]
    pairs = np.full((K, 2, n), -1, np.int32)
    num = np.zeros((K,), np.int32)
    if subm:
        out_shape = list(spatial_shape)
        if n:
            r = lib().sfo_indice_pairs_subm(indices, n, int(batch_size), _i32(out_shape), _i32(ks),
                                            _i32(dl), pairs, num)
            if r < 0:
                raise MemoryError
        return indices.copy(), pairs, num, out_shape
    out_shape = conv_output_size(spatial_shape, ks, st, pd, dl)
    outids = np.zeros((max(n * K, 1), 4), np.int32)
    m = 0
    if n:
        m = lib().sfo_indice_pairs_conv(indices, n, int(batch_size), _i32(out_shape), _i32(ks),
                                        _i32(st), _i32(pd), _i32(dl), outids, pairs, num)
        if m < 0:
            raise MemoryError
    return outids[:m].copy(), pairs, num, out_shape


def linear_index(indices, shape):
    i = indices.astype(np.int64)
    return ((i[:, 0] * shape[0] + i[:, 1]) * shape[1] + i[:, 2]) * shape[2] + i[:, 3]


def canonical_rulebook(in_indices, in_shape, outids, out_shape, pairs, num):
    """SURVEY 8a-R canonical form: outputs sorted by linear index; per offset the set of
    (in_linear, out_linear) pairs sorted by in_linear.  Order-free, row-numbering-free."""
    lin_in = linear_index(in_indices, in_shape)
    lin_out = linear_index(outids, out_shape)
    res = {'out_sorted': np.sort(lin_out), 'num': num.astype(np.int32).copy(), 'pairs': []}
    for k in range(pairs.shape[0]):
        c = int(num[k])
        a = lin_in[pairs[k, 0, :c]]
        b = lin_out[pairs[k, 1, :c]]
        o = np.argsort(a, kind='stable')
        res['pairs'].append(np.stack([a[o], b[o]], 0))
    return res


# ---------------------------------------------------------------- conv

def conv_stage_times(reset=True):
    """(gather, GEMM, scatter-add) seconds spent inside indice_conv since the last reset."""
    buf = (C.c_double * 3)()
    lib().sfo_conv_stage_times(buf, int(bool(reset)))
    return tuple(float(v) for v in buf)


def indice_conv(features, filters_kio, pairs, num, n_out, subm):
    """filters_kio: (K, Cin, Cout) fp32 (spconv-1.x layout flattened)."""
    features = _f32(features)
    w = _f32(filters_kio)
    K, cin, cout = w.shape
    out = np.zeros((n_out, cout), np.float32)
    if n_out == 0:
        return out
    pairs = _i32(pairs)
    r = lib().sfo_indice_conv(features, features.shape[0], cin, w, cout, pairs, _i32(num), K,
                              pairs.shape[2], n_out, int(bool(subm)), out)
    if r < 0:
        raise MemoryError
    return out


def bn_act(x, bn, eps, residual=None, relu=True):
    """bn = dict(weight, bias, running_mean, running_var) -> in-place on a copy."""
    x = _f32(x).copy()
    if x.shape[0] == 0:
        return x
    res = None
    if residual is not None:
        residual = _f32(residual)
        res = residual.ctypes.data_as(C.c_void_p)
    lib().sfo_bn_act(x, x.shape[0], x.shape[1], _f32(bn['weight']), _f32(bn['bias']),
                     _f32(bn['running_mean']), _f32(bn['running_var']), float(eps), res,
                     int(bool(relu)))
    return x


def dense_bev(features, indices, batch, shape):
    features = _f32(features)
    c = features.shape[1]
    D, H, W = shape
    out = np.zeros((batch, c * D, H, W), np.float32)
    lib().sfo_dense_bev(features, _i32(indices), features.shape[0], c, batch, D, H, W, out)
    return out


def dynamic_scatter(feats, coors, reduce_type='max'):
    """dynamic_point_to_voxel_forward (scatter_points_cuda.cu:230-311):
    -> voxel_feats (M,C), voxel_coors (M,ndim), point2voxel (N,), count (M,)."""
    feats, coors = _f32(feats), _i32(coors)
    n, c = feats.shape
    ndim = coors.shape[1]
    vf = np.zeros((max(n, 1), c), np.float32)
    vc = np.zeros((max(n, 1), ndim), np.int32)
    p2v = np.zeros((max(n, 1),), np.int32)
    cnt = np.zeros((max(n, 1),), np.int32)
    m = lib().sfo_dynamic_scatter(feats, coors, n, c, ndim, {'sum': 0, 'mean': 1, 'max': 2}[reduce_type],
                                  vf, vc, p2v, cnt)
    return vf[:m], vc[:m], p2v[:n], cnt[:m]


def weight_to_kio(w, ksize, in_channels, out_channels):
    """Accept spconv-2 layout (Cout,kD,kH,kW,Cin) [write_spconv2.py:46-47] or spconv-1 layout
    (kD,kH,kW,Cin,Cout) [conv.py:98-99]; return (K,Cin,Cout).  When both layouts fit the shape
    (all five extents equal) the spconv-2 reading wins: that is what the reference's
    checkpoints hold."""
    w = np.asarray(w, np.float32)
    ks = tuple(_triple(ksize))
    if w.shape == (out_channels,) + ks + (in_channels,):
        w = np.transpose(w, (1, 2, 3, 4, 0))
    else:
        assert w.shape == ks + (in_channels, out_channels), w.shape
    return np.ascontiguousarray(w.reshape(-1, in_channels, out_channels))


# ---------------------------------------------------------------- SparseEncoder wiring

def encoder_layer_specs(in_channels, base_channels=16, output_channels=128,
                        encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                        encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)),
                        block_type='conv_module'):
    """Flat op list restating sparse_encoder.py:59-93 (post-activate order) and :129-203.

    Each entry: dict(kind='convmodule'|'basicblock', prefix=<state-dict prefix>, cin, cout,
    ksize, stride, padding, subm, indice_key)."""
    specs = [dict(kind='convmodule', prefix='conv_input', cin=in_channels, cout=base_channels,
                  ksize=3, stride=1, padding=1, subm=True, indice_key='subm1')]
    cin = base_channels
    for i, blocks in enumerate(encoder_channels):
        for j, cout in enumerate(tuple(blocks)):
            padding = tuple(encoder_paddings[i])[j]
            prefix = f'encoder_layers.encoder_layer{i + 1}.{j}'
            if i != 0 and j == 0 and block_type == 'conv_module':
                specs.append(dict(kind='convmodule', prefix=prefix, cin=cin, cout=cout, ksize=3,
                                  stride=2, padding=padding, subm=False,
                                  indice_key=f'spconv{i + 1}'))
            elif block_type == 'basicblock':
                if j == len(blocks) - 1 and i != len(encoder_channels) - 1:
                    specs.append(dict(kind='convmodule', prefix=prefix, cin=cin, cout=cout,
                                      ksize=3, stride=2, padding=padding, subm=False,
                                      indice_key=f'spconv{i + 1}'))
                else:
                    specs.append(dict(kind='basicblock', prefix=prefix, cin=cout, cout=cout,
                                      ksize=3, stride=1, padding=1, subm=True, indice_key=None))
            else:
                specs.append(dict(kind='convmodule', prefix=prefix, cin=cin, cout=cout, ksize=3,
                                  stride=1, padding=padding, subm=True,
                                  indice_key=f'subm{i + 1}'))
            cin = cout
    specs.append(dict(kind='convmodule', prefix='conv_out', cin=cin, cout=output_channels,
                      ksize=(3, 1, 1), stride=(2, 1, 1), padding=0, subm=False,
                      indice_key='spconv_down2'))
    return specs


def _bn(sd, prefix):
    return {k: np.asarray(sd[f'{prefix}.{k}'], np.float32)
            for k in ('weight', 'bias', 'running_mean', 'running_var')}


def round_bf16(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32: what v_cvt_pk_bf16_f32 / a torch
    `.bfloat16()` cast does.  Finite inputs only."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = (u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def sparse_encoder_forward(state_dict, voxel_features, coors, batch_size, sparse_shape,
                           in_channels, bn_eps=1e-3, trace=None, bf16=False, **cfg):
    """SparseEncoder.forward (sparse_encoder.py:95-127) on the CPU oracle.

    state_dict: numpy arrays under the reference's key names.  `trace`, if a dict, receives
    per-layer rulebooks / features for parity tests.

    bf16=True restates WHERE the bf16 path of the HIP library rounds (the reference's analogue is
    auto_fp16, sparse_encoder.py:53,95): the first layer runs in fp32, its output and the output
    of every later layer but the last are rounded to bf16, and the filters of every layer but the
    first are rounded to bf16; sums, BatchNorm, residual add and ReLU stay fp32."""
    specs = encoder_layer_specs(in_channels, **cfg)
    n_conv = [0]
    total_conv = sum(1 if sp['kind'] == 'convmodule' else 2 for sp in specs)

    def wq(w):            # filters of the layer about to run
        n_conv[0] += 1
        return round_bf16(w) if (bf16 and n_conv[0] > 1) else w

    def aq(y):            # activations the layer that just ran writes
        return round_bf16(y) if (bf16 and n_conv[0] < total_conv) else y
    feats = _f32(voxel_features)
    idx = _i32(coors)
    shape = list(sparse_shape)
    cache = {}

    def rulebook(spec, idx, shape):
        key = spec['indice_key']
        if key is not None and key in cache:
            return cache[key]
        rb = get_indice_pairs(idx, batch_size, shape, spec['ksize'], spec['stride'],
                              spec['padding'], 1, spec['subm'])
        if key is not None:
            cache[key] = rb
        return rb

    for li, spec in enumerate(specs):
        p = spec['prefix']
        if spec['kind'] == 'convmodule':
            outids, pairs, num, oshape = rulebook(spec, idx, shape)
            w = wq(weight_to_kio(state_dict[f'{p}.0.weight'], spec['ksize'], spec['cin'],
                                 spec['cout']))
            y = indice_conv(feats, w, pairs, num, outids.shape[0], spec['subm'])
            y = aq(bn_act(y, _bn(state_dict, f'{p}.1'), bn_eps, None, True))
            if trace is not None:
                trace[p] = dict(in_indices=idx, in_shape=list(shape), outids=outids,
                                out_shape=list(oshape), pairs=pairs, num=num, features=y)
            feats, idx, shape = y, outids, list(oshape)
        else:  # SparseBasicBlock.forward, sparse_block.py:116-139
            identity = feats
            outids, pairs, num, oshape = rulebook(spec, idx, shape)
            w1 = wq(weight_to_kio(state_dict[f'{p}.conv1.weight'], 3, spec['cin'], spec['cout']))
            y = indice_conv(feats, w1, pairs, num, outids.shape[0], True)
            y = aq(bn_act(y, _bn(state_dict, f'{p}.bn1'), bn_eps, None, True))
            outids2, pairs2, num2, _ = rulebook(spec, idx, shape)  # rebuilt: no indice_key
            w2 = wq(weight_to_kio(state_dict[f'{p}.conv2.weight'], 3, spec['cout'], spec['cout']))
            y = indice_conv(y, w2, pairs2, num2, outids2.shape[0], True)
            y = aq(bn_act(y, _bn(state_dict, f'{p}.bn2'), bn_eps, identity, True))
            if trace is not None:
                trace[p] = dict(in_indices=idx, in_shape=list(shape), outids=outids,
                                out_shape=list(oshape), pairs=pairs, num=num, features=y)
            feats = y
    bev = dense_bev(feats, idx, int(batch_size), shape)
    if trace is not None:
        trace['out_indices'] = idx
        trace['out_shape'] = list(shape)
    return bev


# ---------------------------------------------------------------- SparseUNet wiring

def indice_inverse_conv(features, filters_kio, pairs, num, n_out):
    """Fsp.indice_inverse_conv (ops/spconv/functional.py:39-55 -> indiceConv with inverse = 1,
    spconv_ops.h:319-348): rows are gathered through indicePairs[k][1] and scattered through
    indicePairs[k][0], filter k unchanged -- the forward loop on the swapped pair columns."""
    return indice_conv(features, filters_kio, np.ascontiguousarray(_i32(pairs)[:, ::-1, :]), num,
                       n_out, False)


def sparse_unet_forward(sd, voxel_features, coors, batch_size, sparse_shape, in_channels,
                        base_channels=16, output_channels=128,
                        encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                        encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)),
                        decoder_channels=((64, 64, 64), (64, 64, 32), (32, 32, 16), (16, 16, 16)),
                        decoder_paddings=((1, 0), (1, 0), (0, 0), (0, 1)), bn_eps=1e-3):
    """SparseUNet.forward (mmdet3d/models/middle_encoders/sparse_unet.py:96-169) on the CPU oracle,
    post-activate order.  Returns (spatial_features, seg_features, seg_indices)."""
    cache = {}

    def rulebook(key, idx, shape, ksize, stride, padding, subm):
        if key not in cache:   # conv.py:147-172: keyed rulebooks are built once and reused
            outids, pairs, num, oshape = get_indice_pairs(idx, batch_size, shape, ksize, stride,
                                                          padding, 1, subm)
            cache[key] = (outids, idx, pairs, num, list(shape), list(oshape))
        return cache[key]

    def convmodule(prefix, feats, idx, shape, cin, cout, ksize, stride, padding, subm, key,
                   inverse=False):
        w = weight_to_kio(sd[f'{prefix}.0.weight'], ksize, cin, cout)
        if inverse:
            _, in_idx, pairs, num, in_shape, _ = cache[key]
            y = indice_inverse_conv(feats, w, pairs, num, in_idx.shape[0])
            idx, shape = in_idx, in_shape
        else:
            outids, _, pairs, num, _, oshape = rulebook(key, idx, shape, ksize, stride, padding, subm)
            y = indice_conv(feats, w, pairs, num, outids.shape[0], subm)
            idx, shape = outids, oshape
        return bn_act(y, _bn(sd, f'{prefix}.1'), bn_eps, None, True), idx, list(shape)

    feats, idx, shape = _f32(voxel_features), _i32(coors), list(sparse_shape)
    feats, idx, shape = convmodule('conv_input', feats, idx, shape, in_channels, base_channels, 3, 1,
                                   1, True, 'subm1')
    cin = base_channels
    enc = []
    for i, blocks in enumerate(encoder_channels):                          # :191-235
        for j, cout in enumerate(tuple(blocks)):
            p = f'encoder_layers.encoder_layer{i + 1}.{j}'
            pad = tuple(encoder_paddings[i])[j]
            if i != 0 and j == 0:
                feats, idx, shape = convmodule(p, feats, idx, shape, cin, cout, 3, 2, pad, False,
                                               f'spconv{i + 1}')
            else:
                feats, idx, shape = convmodule(p, feats, idx, shape, cin, cout, 3, 1, pad, True,
                                               f'subm{i + 1}')
            cin = cout
        enc.append((feats, idx, shape))
    y, oidx, oshape = convmodule('conv_out', feats, idx, shape, cin, output_channels, (3, 1, 1),
                                 (2, 1, 1), 0, False, 'spconv_down2')
    spatial = dense_bev(y, oidx, int(batch_size), oshape)
    stages = len(decoder_channels)
    bottom = enc[-1][0]
    for i, (c_lat, c_merge, c_up) in enumerate(decoder_channels):          # :133-169, :237-292
        level = stages - i
        lat, lidx, lshape = enc[level - 1]
        outids, _, pairs, num, _, _ = rulebook(f'subm{level}', lidx, lshape, 3, 1, 1, True)
        p = f'lateral_layer{level}'                                         # SparseBasicBlock
        h = indice_conv(lat, weight_to_kio(sd[f'{p}.conv1.weight'], 3, cin, c_lat), pairs, num,
                        outids.shape[0], True)
        h = bn_act(h, _bn(sd, f'{p}.bn1'), bn_eps, None, True)
        h = indice_conv(h, weight_to_kio(sd[f'{p}.conv2.weight'], 3, c_lat, c_lat), pairs, num,
                        outids.shape[0], True)
        h = bn_act(h, _bn(sd, f'{p}.bn2'), bn_eps, lat, True)
        cat = np.concatenate([bottom, h], axis=1)                           # :164
        merged, _, _ = convmodule(f'merge_layer{level}', cat, lidx, lshape, cin * 2, c_merge, 3, 1,
                                  decoder_paddings[i][0], True, f'subm{level}')
        red = cat.reshape(cat.shape[0], c_merge, -1).sum(axis=2)            # reduce_channel :171-189
        x = (merged + red).astype(np.float32)                               # :167
        if level != 1:
            bottom, idx, shape = convmodule(f'upsample_layer{level}', x, lidx, lshape, cin, c_up, 3,
                                            1, 0, False, f'spconv{level}', inverse=True)
        else:
            bottom, idx, shape = convmodule(f'upsample_layer{level}', x, lidx, lshape, cin, c_up, 3,
                                            1, decoder_paddings[i][1], True, 'subm1')
        cin = c_up
    return spatial, bottom, idx


# ------------------------------------------------------------------ BEV backbone / neck
# The reference's SECOND / SECONDFPN are chains of torch modules (nn.Conv2d, nn.ConvTranspose2d,
# nn.BatchNorm2d(eps=1e-3), nn.ReLU), so the reference's CPU arithmetic IS torch's CPU kernels: the
# oracle calls the same functional ops in fp32 on the host.  tests/test_oracle_second.py pins this
# transcription against a direct numpy statement of the two convolutions on small shapes.

def _bn2d(sd, prefix):
    return {k: np.asarray(sd[f'{prefix}.{k}'], np.float32)
            for k in ('weight', 'bias', 'running_mean', 'running_var')}


def conv2d_bn_relu(x, weight, bn, stride, padding, eps=1e-3, bias=None, transposed=False):
    """relu(BatchNorm2d_eval(conv(x))): one conv -> norm -> ReLU triple of
    mmdet3d/models/backbones/second.py:37-57 / necks/second_fpn.py:61-63."""
    import torch
    import torch.nn.functional as F
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32))
    with torch.no_grad():
        if transposed:
            y = F.conv_transpose2d(t(x), t(weight), t(bias), stride=stride)
        else:
            y = F.conv2d(t(x), t(weight), t(bias), stride=stride, padding=padding)
        y = F.batch_norm(y, t(bn['running_mean']), t(bn['running_var']), t(bn['weight']),
                         t(bn['bias']), False, 0.0, eps)
        return F.relu(y).numpy()


def second_forward(sd, x, layer_nums=(3, 5, 5), layer_strides=(2, 2, 2), eps=1e-3):
    """SECOND.forward (second.py:73-86) with the block layout of second.py:36-60: state-dict keys
    blocks.I.{0,3,6,...}.weight (conv) and blocks.I.{1,4,7,...}.* (BatchNorm2d)."""
    outs = []
    for i, (n, s) in enumerate(zip(layer_nums, layer_strides)):
        for j in range(n + 1):
            x = conv2d_bn_relu(x, sd[f'blocks.{i}.{3 * j}.weight'], _bn2d(sd, f'blocks.{i}.{3 * j + 1}'),
                               s if j == 0 else 1, 1, eps, sd.get(f'blocks.{i}.{3 * j}.bias'))
        outs.append(x)
    return tuple(outs)


def second_fpn_forward(sd, xs, upsample_strides=(1, 2, 4), use_conv_for_no_stride=False, eps=1e-3):
    """SECONDFPN.forward (second_fpn.py:76-92): deblock i = deconv(k = s = stride) when stride > 1 or
    (stride == 1 and not use_conv_for_no_stride), else conv(k = s = round(1/stride))
    (second_fpn.py:43-59); outputs concatenated along channels."""
    ups = []
    for i, (x, stride) in enumerate(zip(xs, upsample_strides)):
        w, bn = sd[f'deblocks.{i}.0.weight'], _bn2d(sd, f'deblocks.{i}.1')
        b = sd.get(f'deblocks.{i}.0.bias')
        if stride > 1 or (stride == 1 and not use_conv_for_no_stride):
            ups.append(conv2d_bn_relu(x, w, bn, stride, 0, eps, b, transposed=True))
        else:
            s = int(np.round(1 / stride))
            ups.append(conv2d_bn_relu(x, w, bn, s, 0, eps, b))
    return [np.concatenate(ups, axis=1) if len(ups) > 1 else ups[0]]

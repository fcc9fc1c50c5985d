"""ctypes binding of libsparsefusion_hip.so (the C ABI declared in include/sparsefusion_hip.h).

There is no fallback: if the library is missing or a tensor is not on a HIP device the ops raise.
PyTorch is used only for device memory and streams.
"""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libsparsefusion_hip.so')
_LIB = None

SF_OK = 0

_vp, _i, _sz, _ll = C.c_void_p, C.c_int, C.c_size_t, C.c_longlong
_F3, _F6, _I3 = C.c_float * 3, C.c_float * 6, C.c_int * 3   # array instances decay to pointers
_f3 = _f6 = C.POINTER(C.c_float)
_i3 = C.POINTER(C.c_int)


class SfCoordSet(C.Structure):
    _fields_ = [('shape', C.c_int * 3), ('bound', C.c_int), ('indices', C.c_void_p),
                ('keys', C.c_void_p), ('vals', C.c_void_p), ('capacity', C.c_int)]


class SfRulebook(C.Structure):
    _fields_ = [('src', C.c_int), ('dst', C.c_int), ('ksize', C.c_int * 3), ('stride', C.c_int * 3),
                ('padding', C.c_int * 3), ('subm', C.c_int), ('nbr', C.c_void_p),
                ('nbr_stride', C.c_int), ('plan', C.c_void_p), ('plan_ws', C.c_void_p),
                ('plan_ws_bytes', C.c_size_t), ('first', C.c_void_p)]


class SfFrontDesc(C.Structure):
    _fields_ = [('batch', C.c_int), ('points', C.POINTER(C.c_void_p)), ('num_points', C.POINTER(C.c_int)),
                ('num_features', C.c_int), ('voxel_size', C.c_float * 3), ('coors_range', C.c_float * 6),
                ('max_points', C.c_int), ('max_voxels', C.c_int), ('voxels', C.c_void_p),
                ('num_points_per_voxel', C.c_void_p), ('mean', C.c_void_p), ('mean_features', C.c_int),
                ('mean_stride', C.c_int), ('vox_workspace', C.c_void_p), ('vox_workspace_bytes', C.c_size_t),
                ('n_sets', C.c_int), ('sets', C.POINTER(SfCoordSet)), ('counts_dev', C.c_void_p),
                ('n_rulebooks', C.c_int), ('rulebooks', C.POINTER(SfRulebook)),
                ('rb_workspace', C.c_void_p), ('rb_workspace_bytes', C.c_size_t)]


# name -> (restype, argtypes); kept in step with include/sparsefusion_hip.h (tests check it)
SIGNATURES = {
    'sf_version': (_i, []),
    'sf_last_error': (C.c_char_p, []),
    'sf_dynamic_voxelize': (_i, [_vp, _i, _i, _f3, _f6, _vp, _vp]),
    'sf_hard_voxelize_workspace': (_sz, [_i, _i]),
    'sf_hard_voxelize': (_i, [_vp, _i, _i, _f3, _f6, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _i,
                              _vp, _i, _vp, C.POINTER(_i), _vp, _sz, _vp]),
    'sf_hard_voxelize_static': (_i, [_vp, _i, _i, _f3, _f6, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i,
                                     _i, _vp, _i, _vp, _vp]),
    'sf_voxel_mean': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    'sf_vfe_layer': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _f3, _f6, _vp, _i, _vp, _vp, _i, _vp,
                          _vp]),
    'sf_voxel_heights': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'sf_hash_capacity': (_i, [_ll]),
    'sf_hash_build': (_i, [_vp, _vp, _i, _i3, _i, _vp, _vp, _i, _vp, _vp]),
    'sf_conv_output_set_workspace': (_sz, [_i, _i]),
    'sf_conv_output_set': (_i, [_vp, _vp, _i, _i, _i3, _i3, _i3, _i3, _vp, _vp, _i, _vp, _i, _vp,
                                _vp, _vp, _sz, _vp]),
    'sf_strided_rulebook_workspace': (_sz, [_i, _i, _i3, _i3]),
    'sf_strided_rulebook': (_i, [_vp, _vp, _i, _i, _i3, _i3, _i3, _i3, _vp, _vp, _i, _vp, _i, _vp,
                                 _vp, _vp, _i, _vp, _sz, _vp]),
    'sf_neighbor_table': (_i, [_vp, _vp, _i, _i, _i3, _i3, _i3, _i3, _vp, _vp, _i, _i, _vp, _i,
                               _vp]),
    'sf_export_indice_pairs_workspace': (_sz, [_i, _i]),
    'sf_export_indice_pairs': (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'sf_indice_pairs_to_table': (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    'sf_lidar_front': (_i, [C.POINTER(SfFrontDesc), _vp]),
    'sf_conv_padded_cin': (_i, [_i]),
    'sf_pack_filters_f32': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'sf_sparse_conv_f32': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp,
                                _i, _vp]),
    'sf_last_conv_kernel': (C.c_char_p, []),
    'sf_row_order_workspace': (_sz, [_i]),
    'sf_row_order_size': (_i, [_i]),
    'sf_row_order_by_mask': (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    'sf_sparse_conv_f32_ordered': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i,
                                        _vp, _vp, _i, _vp]),
    'sf_conv_padded_cin_bf16': (_i, [_i]),
    'sf_pack_filters_bf16': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'sf_cast_f32_bf16': (_i, [_vp, _i, _vp, _i, _vp, _i, _vp]),
    'sf_sparse_conv_bf16': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp,
                                 _i, _vp, _i, _vp]),
    'sf_dense_bev': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'sf_dense_bev_mark': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'sf_dense_bev_write': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'sf_conv2d_packed_bytes': (_sz, [_i, _i, _i, _i]),
    'sf_conv2d_pack_filters_f32': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'sf_conv2d_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                           _vp]),
    'sf_pad_channels_f32': (_i, [_vp, _i, _vp, _i, _vp, _i, _vp]),
    'sf_coors_max': (_i, [_vp, _i, _i, _vp, _vp]),
    'sf_dynamic_scatter_workspace': (_sz, [C.POINTER(C.c_int), _i]),
    'sf_dynamic_scatter': (_i, [_vp, _vp, _i, _i, _i, C.POINTER(C.c_int), _i, _vp, _vp, _vp, _vp,
                                _i, _vp, _vp, _vp, _sz, _vp]),
}


def build(verbose=False):
    """Compile the HIP library in-tree (hipcc --offload-arch=gfx950); works without a GPU."""
    cmd = ['make', '-C', os.path.join(_HERE, 'csrc'), '-j8']
    if not verbose:
        cmd.insert(1, '-s')
    subprocess.check_call(cmd)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; '
                f'g.build()"` (there is no non-HIP fallback)')
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != SF_OK:
        msg = lib().sf_last_error().decode('utf-8', 'replace')
        raise RuntimeError(f'sparsefusion_hip error {rc}: {msg}')


def f3(v):
    return _F3(*[float(x) for x in v])


def f6(v):
    return _F6(*[float(x) for x in v])


def i3(v):
    if not isinstance(v, (list, tuple)):
        v = [v, v, v]
    assert len(v) == 3
    return _I3(*[int(x) for x in v])


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: no CPU path exists."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('sparsefusion_amd ops need tensors on a HIP device '
                           f'(got {t.device}); there is no CPU path')
    if not t.is_contiguous():
        raise RuntimeError('sparsefusion_amd ops need contiguous tensors')
    return t.data_ptr()


def stream(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def require(t, dtype, what):
    if t.dtype != dtype:
        raise RuntimeError(f'{what} must be {dtype}, got {t.dtype}')
    return t


class _ForwardOnly(torch.autograd.Function):
    """Identity whose backward raises: the kernels behind this library are forward-only (the LiDAR
    branch is frozen in the target configs), and an output without a grad_fn would let
    loss.backward() skip these weights without a word."""

    @staticmethod
    def forward(ctx, out, *deps):
        return out.view_as(out)

    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError(
            'sparsefusion_amd kernels are forward-only: freeze these parameters '
            '(requires_grad_(False)) or run under torch.no_grad()')


def forward_only(out, *deps):
    """Tie `out` (a tensor the C ABI wrote) to the tensors it was computed from, so that autograd
    reaches a backward that raises NotImplementedError instead of silently passing them by."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in deps):
        return _ForwardOnly.apply(out, *deps)
    return out

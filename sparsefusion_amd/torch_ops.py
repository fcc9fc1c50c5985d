"""PyTorch-ROCm custom ops over the C ABI: `torch.ops.sparsefusion.*`.

The reference reaches its native code through two pybind11 torch extensions, `voxel_layer`
(mmdet3d/ops/voxel/src/voxelization.cpp:6-11) and `sparse_conv_ext`
(mmdet3d/ops/spconv/src/all.cc:21-51).  The ops below carry the same names, argument order and
return values as those bindings, registered with `torch.library` for the HIP device key
("CUDA" is the dispatch key PyTorch-ROCm uses), so code written against
`voxel_layer.hard_voxelize(...)` / `sparse_conv_ext.get_indice_pairs_3d(...)` can call
`torch.ops.sparsefusion.hard_voxelize(...)` unchanged.  Each implementation only turns tensors
into pointers and calls libsparsefusion_hip.so; there is no CPU registration (host tensors fail
in the dispatcher, as the rest of the package refuses them).
"""
import torch

from . import scatter_points, spconv, voxelize

_LIB = torch.library.Library('sparsefusion', 'DEF')

_LIB.define('hard_voxelize(Tensor points, Tensor(a!) voxels, Tensor(b!) coors, '
            'Tensor(c!) num_points_per_voxel, float[] voxel_size, float[] coors_range, '
            'int max_points, int max_voxels, int NDim=3) -> int')
_LIB.define('dynamic_voxelize(Tensor points, Tensor(a!) coors, float[] voxel_size, '
            'float[] coors_range, int NDim=3) -> ()')
_LIB.define('dynamic_point_to_voxel_forward(Tensor feats, Tensor coors, str reduce_type) '
            '-> Tensor[]')
_LIB.define('get_indice_pairs_3d(Tensor indices, int batch_size, int[] out_shape, '
            'int[] spatial_shape, int[] ksize, int[] stride, int[] padding, int[] dilation, '
            'int[] out_padding, int subm, int transpose) -> Tensor[]')
_LIB.define('indice_conv_fp32(Tensor features, Tensor filters, Tensor indice_pairs, '
            'Tensor indice_num, int num_act_out, int inverse, int subm) -> Tensor')
_LIB.define('dense_bev(Tensor features, Tensor indices, int batch_size, int[] spatial_shape) '
            '-> Tensor')


def _hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range,
                   max_points, max_voxels, NDim=3):
    return voxelize.hard_voxelize(points, voxels, coors, num_points_per_voxel, list(voxel_size),
                                  list(coors_range), max_points, max_voxels, NDim)


def _dynamic_voxelize(points, coors, voxel_size, coors_range, NDim=3):
    voxelize.dynamic_voxelize(points, coors, list(voxel_size), list(coors_range), NDim)


def _dynamic_point_to_voxel_forward(feats, coors, reduce_type):
    return scatter_points.dynamic_point_to_voxel_forward(feats, coors, reduce_type)


def _get_indice_pairs_3d(indices, batch_size, out_shape, spatial_shape, ksize, stride, padding,
                         dilation, out_padding, subm, transpose):
    """sparse_conv_ext.get_indice_pairs_3d (all.cc:21-33, spconv_ops.h:27-141) ->
    [outids, indice_pairs (K,2,N), indice_num (K)]; out_shape is recomputed and checked."""
    outids, pairs, num = spconv.get_indice_pairs(indices, batch_size, list(spatial_shape),
                                                 list(ksize), list(stride), list(padding),
                                                 list(dilation), list(out_padding), bool(subm),
                                                 bool(transpose))
    if not subm:
        want = spconv.get_conv_output_size(list(spatial_shape), list(ksize), list(stride),
                                           list(padding), list(dilation))
        if list(out_shape) != want:
            raise RuntimeError(f'out_shape {list(out_shape)} does not match the geometry {want}')
    return [outids, pairs, num]


def _indice_conv_fp32(features, filters, indice_pairs, indice_num, num_act_out, inverse, subm):
    return spconv.indice_conv(features, filters, indice_pairs, indice_num, num_act_out,
                              bool(inverse), bool(subm))


def _dense_bev(features, indices, batch_size, spatial_shape):
    """SparseConvTensor(features, indices, spatial_shape, batch_size).dense() viewed as
    (N, C*D, H, W) (structure.py:55-64, sparse_encoder.py:122-125)."""
    d = spconv.SparseConvTensor(features, indices, list(spatial_shape), batch_size).dense()
    n, c, dd, h, w = d.shape
    return d.view(n, c * dd, h, w)


for _name, _fn in (('hard_voxelize', _hard_voxelize), ('dynamic_voxelize', _dynamic_voxelize),
                   ('dynamic_point_to_voxel_forward', _dynamic_point_to_voxel_forward),
                   ('get_indice_pairs_3d', _get_indice_pairs_3d),
                   ('indice_conv_fp32', _indice_conv_fp32), ('dense_bev', _dense_bev)):
    _LIB.impl(_name, _fn, 'CUDA')

OPS = ('hard_voxelize', 'dynamic_voxelize', 'dynamic_point_to_voxel_forward',
       'get_indice_pairs_3d', 'indice_conv_fp32', 'dense_bev')

"""`Voxelization` / `voxelization`: drop-in for mmdet3d/ops/voxel/voxelize.py:10-122 on MI355X.

Same constructor, attributes, forward signature and outputs as the reference module; the native
op behind it is libsparsefusion_hip.so's sf_hard_voxelize / sf_dynamic_voxelize instead of the
`voxel_layer` CUDA extension.  Forward only (the reference op has no backward either: it is
called under torch.no_grad, detectors/sparsefusion.py:133).
"""
import ctypes as C

import torch
from torch import nn
from torch.nn.modules.utils import _pair

from . import _lib


def dynamic_voxelize(points, coors, voxel_size, coors_range, NDim=3):
    """voxel_layer.dynamic_voxelize (voxelization.cpp:8): fills coors (N,3) int32 in place."""
    assert NDim == 3
    _lib.require(points, torch.float32, 'points')
    _lib.require(coors, torch.int32, 'coors')
    _lib.check(_lib.lib().sf_dynamic_voxelize(
        _lib.ptr(points), points.size(0), points.size(1), _lib.f3(voxel_size),
        _lib.f6(coors_range), _lib.ptr(coors), _lib.stream(points.device)))


def hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range,
                  max_points, max_voxels, NDim=3):
    """voxel_layer.hard_voxelize (voxelization.cpp:7): fills the caller-allocated outputs in
    place and returns voxel_num (host int; this is the reference's one host sync)."""
    assert NDim == 3
    L = _lib.lib()
    _lib.require(points, torch.float32, 'points')
    _lib.require(voxels, torch.float32, 'voxels')
    _lib.require(coors, torch.int32, 'coors')
    _lib.require(num_points_per_voxel, torch.int32, 'num_points_per_voxel')
    n, f = points.size(0), points.size(1)
    if voxels.size(0) < max_voxels or voxels.size(1) != max_points or voxels.size(2) != f:
        raise RuntimeError('voxels must be (max_voxels, max_points, num_features)')
    ws_bytes = L.sf_hard_voxelize_workspace(n, max_points)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=points.device)
    count = torch.zeros(2, dtype=torch.int32, device=points.device)  # [row count, overflow]
    vnum = C.c_int(0)
    _lib.check(L.sf_hard_voxelize(
        _lib.ptr(points), n, f, _lib.f3(voxel_size), _lib.f6(coors_range), max_points, max_voxels,
        _lib.ptr(voxels), _lib.ptr(coors), coors.size(1), 0, _lib.ptr(num_points_per_voxel),
        None, 0, 0, count.data_ptr(), min(voxels.size(0), coors.size(0)),
        count.data_ptr() + 4, C.byref(vnum), ws.data_ptr(), ws_bytes, _lib.stream(points.device)))
    return vnum.value


class _Voxelization(torch.autograd.Function):

    @staticmethod
    def forward(ctx, points, voxel_size, coors_range, max_points=35, max_voxels=20000):
        """Same contract as the reference (voxelize.py:13-58)."""
        if not points.is_contiguous():
            points = points.contiguous()
        if max_points == -1 or max_voxels == -1:
            coors = points.new_zeros(size=(points.size(0), 3), dtype=torch.int)
            dynamic_voxelize(points, coors, voxel_size, coors_range, 3)
            return coors
        # The reference zero-fills cap-sized outputs (voxelize.py:46-50); the kernels here write
        # every element of the rows they return, so plain empty allocations suffice.
        voxels = points.new_empty(size=(max_voxels, max_points, points.size(1)))
        coors = points.new_empty(size=(max_voxels, 3), dtype=torch.int)
        num_points_per_voxel = points.new_empty(size=(max_voxels, ), dtype=torch.int)
        voxel_num = hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size,
                                  coors_range, max_points, max_voxels, 3)
        return voxels[:voxel_num], coors[:voxel_num], num_points_per_voxel[:voxel_num]


voxelization = _Voxelization.apply


class Voxelization(nn.Module):
    """voxelize.py:64-122: same arguments, attributes and repr."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.max_num_points = max_num_points
        # (training, testing) caps; a bare int applies to both
        self.max_voxels = max_voxels if isinstance(max_voxels, tuple) else _pair(max_voxels)
        # grid extents in fp32, rounded, exactly as the native op derives them
        pcr = torch.as_tensor(point_cloud_range, dtype=torch.float32)
        extent = (pcr[3:] - pcr[:3]) / torch.as_tensor(voxel_size, dtype=torch.float32)
        self.grid_size = torch.round(extent).long()          # (x, y, z) cell counts
        self.pcd_shape = [1, self.grid_size[1], self.grid_size[0]]  # [d, h, w] of the BEV canvas

    def forward(self, input):
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        return voxelization(input, self.voxel_size, self.point_cloud_range, self.max_num_points,
                            max_voxels)

    def __repr__(self):
        fields = (('voxel_size', self.voxel_size), ('point_cloud_range', self.point_cloud_range),
                  ('max_num_points', self.max_num_points), ('max_voxels', self.max_voxels))
        return f"{type(self).__name__}(" + ', '.join(f'{k}={v}' for k, v in fields) + ')'

"""`DynamicScatter` / `dynamic_scatter`: reduce per-point features into unique voxels.

Mirror of mmdet3d/ops/voxel/scatter_points.py:9-107 (forward only: the LiDAR branch runs frozen,
SURVEY 3.4).  Output voxels come in ascending linearised-coordinate order, points with a negative
coordinate are dropped, as dynamic_point_to_voxel_forward_gpu does (scatter_points_cuda.cu:230-311).
Like the reference there is no CPU path (voxelization.h:106 "do not support cpu yet").
"""
import ctypes as C

import torch
from torch import nn

from . import _lib

_REDUCE = {'sum': 0, 'mean': 1, 'max': 2}   # reduce_t, scatter_points_cuda.cu:7


def dynamic_point_to_voxel_forward(feats, coors, reduce_type='max'):
    """-> [voxel_feats (M,C), voxel_coors (M,ndim), point2voxel_map (N,), voxel_points_count (M,)],
    the four tensors of the reference's native op (voxelization.h:96-108)."""
    if reduce_type not in _REDUCE:
        raise RuntimeError('do not support reduce type ' + str(reduce_type))
    _lib.require(feats, torch.float32, 'feats')
    _lib.require(coors, torch.int32, 'coors')
    if feats.dim() != 2 or coors.dim() != 2 or feats.shape[0] != coors.shape[0]:
        raise RuntimeError('feats must be (N, C) and coors (N, ndim)')
    L = _lib.lib()
    dev = feats.device
    n, c = feats.shape
    ndim = coors.shape[1]
    fp, cp = _lib.ptr(feats), _lib.ptr(coors)   # refuses host tensors: no CPU path
    st = _lib.stream(dev)
    # size of the linear-id space: per-dimension maximum + 1 (scatter_points_cuda.cu:241)
    mx = torch.empty(ndim, dtype=torch.int32, device=dev)
    _lib.check(L.sf_coors_max(cp, n, ndim, mx.data_ptr(), st))
    dims = [max(int(v) + 1, 1) for v in mx.tolist()]           # host sync (the reference syncs too)
    cdims = (C.c_int * ndim)(*dims)
    ws_bytes = L.sf_dynamic_scatter_workspace(cdims, ndim)
    if ws_bytes == 0:
        raise RuntimeError(f'coordinate space {dims} has more than 2^32 cells')
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    cells = 1
    for d in dims:
        cells *= d
    cap = min(n, cells)
    vf = torch.empty((cap, c), dtype=torch.float32, device=dev)
    vc = torch.empty((cap, ndim), dtype=torch.int32, device=dev)
    cnt = torch.empty((cap,), dtype=torch.int32, device=dev)
    p2v = torch.empty((n,), dtype=torch.int32, device=dev)
    meta = torch.zeros(2, dtype=torch.int32, device=dev)        # [num voxels, overflow]
    _lib.check(L.sf_dynamic_scatter(fp, cp, n, c, ndim, cdims, _REDUCE[reduce_type], vf.data_ptr(),
                                    vc.data_ptr(), p2v.data_ptr(), cnt.data_ptr(), cap,
                                    meta.data_ptr(), meta.data_ptr() + 4, ws.data_ptr(), ws_bytes,
                                    st))
    m, ovf = meta.tolist()
    if ovf:
        raise RuntimeError('dynamic scatter output capacity exceeded')
    return [vf[:m], vc[:m], p2v, cnt[:m]]


def dynamic_scatter(feats, coors, reduce_type='max'):
    """(voxel_feats, voxel_coors), as `_dynamic_scatter.apply` (scatter_points.py:12-34)."""
    vf, vc, _, _ = dynamic_point_to_voxel_forward(feats, coors, reduce_type)
    return vf, vc


class DynamicScatter(nn.Module):
    """scatter_points.py:53-107: same constructor, attributes, forward and repr."""

    def __init__(self, voxel_size, point_cloud_range, average_points: bool):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.average_points = average_points

    def forward_single(self, points, coors):
        reduce = 'mean' if self.average_points else 'max'
        return dynamic_scatter(points.contiguous(), coors.contiguous(), reduce)

    def forward(self, points, coors):
        if coors.size(-1) == 3:
            return self.forward_single(points, coors)
        batch_size = int(coors[-1, 0]) + 1
        voxels, voxel_coors = [], []
        for i in range(batch_size):
            inds = torch.where(coors[:, 0] == i)
            voxel, voxel_coor = self.forward_single(points[inds], coors[inds][:, 1:])
            voxel_coors.append(nn.functional.pad(voxel_coor, (1, 0), mode='constant', value=i))
            voxels.append(voxel)
        return torch.cat(voxels, dim=0), torch.cat(voxel_coors, dim=0)

    def __repr__(self):
        return (f'{self.__class__.__name__}(voxel_size={self.voxel_size}, '
                f'point_cloud_range={self.point_cloud_range}, '
                f'average_points={self.average_points})')

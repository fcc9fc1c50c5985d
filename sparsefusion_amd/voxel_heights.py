"""`extract_voxel_heights`: per-BEV-cell min / max point height (SparseFusion-specific).

Mirror of SparseFusionDetector.extract_voxel_heights (mmdet3d/models/detectors/sparsefusion.py:73-112),
which takes `grid_size` and `out_size_factor` from `test_cfg['pts']`; here they are arguments.
"""
import torch

from . import _lib


def extract_voxel_heights(voxels, coors, grid_size, out_size_factor, batch_size=None):
    """voxels (M, max_points, F>=3) f32, coors (M, 4) i32 (b, z, y, x) ->
    (min_voxel_height, max_voxel_height), each (B, grid_y // osf, grid_x // osf) f32; empty cells
    hold +100 / -200 (sparsefusion.py:87-88).  batch_size defaults to coors[-1, 0] + 1 (:74)."""
    _lib.require(voxels, torch.float32, 'voxels')
    _lib.require(coors, torch.int32, 'coors')
    if voxels.dim() != 3 or coors.dim() != 2 or coors.shape[1] != 4 or coors.shape[0] != voxels.shape[0]:
        raise RuntimeError('voxels must be (M, P, F) and coors (M, 4)')
    vp, cp = _lib.ptr(voxels), _lib.ptr(coors)
    dev = voxels.device
    m, mp, f = voxels.shape
    if batch_size is None:
        batch_size = int(coors[-1, 0].item()) + 1 if m else 1
    osf = int(out_size_factor)
    nx, ny = int(grid_size[0]) // osf, int(grid_size[1]) // osf
    mn = torch.empty((batch_size, ny, nx), dtype=torch.float32, device=dev)
    mx = torch.empty_like(mn)
    _lib.check(_lib.lib().sf_voxel_heights(vp, cp, None, m, mp, f, batch_size, ny, nx, osf,
                                           mn.data_ptr(), mx.data_ptr(), _lib.stream(dev)))
    return mn, mx

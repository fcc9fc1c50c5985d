"""`PointPillarsScatter`: drop-in for mmdet3d/models/middle_encoders/pillar_scatter.py:8-101
(SURVEY 8f row 4).  The pillar -> pseudo-image scatter is the D = 1 case of the dense BEV writer
(sf_dense_bev): one coalesced write of the (B, C, ny, nx) canvas, zeros included."""
import torch
from torch import nn

from . import _lib
from .registry import MIDDLE_ENCODERS


@MIDDLE_ENCODERS.register_module()
class PointPillarsScatter(nn.Module):

    def __init__(self, in_channels, output_shape):
        super().__init__()
        self.output_shape = output_shape
        self.ny = output_shape[0]
        self.nx = output_shape[1]
        self.in_channels = in_channels
        self.fp16_enabled = False

    def _scatter(self, voxel_features, b, y, x, batch_size):
        _lib.require(voxel_features, torch.float32, 'voxel_features')
        n, c = voxel_features.shape
        if c != self.in_channels:
            raise RuntimeError(f'expected {self.in_channels} channels, got {c}')
        idx = torch.zeros((n, 4), dtype=torch.int32, device=voxel_features.device)   # (b, 0, y, x)
        idx[:, 0], idx[:, 2], idx[:, 3] = b, y, x
        out = torch.empty((batch_size, c, self.ny, self.nx), dtype=torch.float32,
                          device=voxel_features.device)
        _lib.check(_lib.lib().sf_dense_bev(
            _lib.ptr(voxel_features.contiguous()), _lib.ptr(idx), None, n, c, batch_size, 1, self.ny,
            self.nx, out.data_ptr(), _lib.stream(voxel_features.device)))
        return out

    def forward(self, voxel_features, coors, batch_size=None):
        if batch_size is not None:
            return self.forward_batch(voxel_features, coors, batch_size)
        return self.forward_single(voxel_features, coors)

    def forward_single(self, voxel_features, coors):
        """pillar_scatter.py:37-60: one sample, cell = coors[:, 1] * nx + coors[:, 2]; returns a
        list holding the (1, C, ny, nx) canvas."""
        return [self._scatter(voxel_features, 0, coors[:, 1].int(), coors[:, 2].int(), 1)]

    def forward_batch(self, voxel_features, coors, batch_size):
        """pillar_scatter.py:62-101: coors (N, 4) = (sample, z, y, x), cell = y * nx + x."""
        return self._scatter(voxel_features, coors[:, 0].int(), coors[:, 2].int(),
                             coors[:, 3].int(), int(batch_size))

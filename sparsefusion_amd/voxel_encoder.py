"""`HardSimpleVFE`: drop-in for mmdet3d/models/voxel_encoders/voxel_encoder.py:12-44."""
import torch
from torch import nn

from . import _lib
from .registry import VOXEL_ENCODERS


@VOXEL_ENCODERS.register_module()
class HardSimpleVFE(nn.Module):
    """Mean of the points in each voxel (the SECOND voxel feature encoder)."""

    def __init__(self, num_features=4):
        super().__init__()
        self.num_features = num_features
        self.fp16_enabled = False

    def forward(self, features, num_points, coors=None):
        """features (M, max_points, F) f32, num_points (M,) -> (M, num_features) contiguous.
        Sum over every slot (zero padded ones included) divided by num_points, as the
        reference's `features[:, :, :nf].sum(dim=1) / num_points` (voxel_encoder.py:42-44)."""
        _lib.require(features, torch.float32, 'features')
        m, mp, f = features.shape
        out = torch.empty((m, self.num_features), dtype=torch.float32, device=features.device)
        _lib.check(_lib.lib().sf_voxel_mean(
            _lib.ptr(features.contiguous()), _lib.ptr(num_points.int().contiguous()), None, m, mp,
            f, self.num_features, out.data_ptr(), self.num_features,
            _lib.stream(features.device)))
        return out

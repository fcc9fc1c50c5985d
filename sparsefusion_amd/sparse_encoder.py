"""`SparseEncoder`: drop-in for mmdet3d/models/middle_encoders/sparse_encoder.py:9-203.

Same constructor arguments, module tree and state-dict keys; forward returns the
dense BEV map (N, C*D, H, W).
"""
import torch
from torch import nn

from . import spconv
from .registry import MIDDLE_ENCODERS
from .sparse_block import SparseBasicBlock, make_sparse_convmodule


@MIDDLE_ENCODERS.register_module()
class SparseEncoder(nn.Module):

    def __init__(self,
                 in_channels,
                 sparse_shape,
                 order=('conv', 'norm', 'act'),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01),
                 base_channels=16,
                 output_channels=128,
                 encoder_channels=((16, ), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1, ), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)),
                 block_type='conv_module'):
        super().__init__()
        assert block_type in ['conv_module', 'basicblock']
        assert isinstance(order, tuple) and len(order) == 3
        assert set(order) == {'conv', 'norm', 'act'}
        self.sparse_shape = sparse_shape
        self.in_channels = in_channels
        self.order = order
        self.base_channels = base_channels
        self.output_channels = output_channels
        self.encoder_channels = encoder_channels
        self.encoder_paddings = encoder_paddings
        self.stage_num = len(self.encoder_channels)
        self.fp16_enabled = False

        # pre-activation orders start with a bare conv (sparse_encoder.py:59-68)
        input_order = ('conv', ) if self.order[0] != 'conv' else ('conv', 'norm', 'act')
        self.conv_input = make_sparse_convmodule(
            in_channels, self.base_channels, 3, norm_cfg=norm_cfg, padding=1, indice_key='subm1',
            conv_type='SubMConv3d', order=input_order)
        encoder_out_channels = self.make_encoder_layers(
            make_sparse_convmodule, norm_cfg, self.base_channels, block_type=block_type)
        self.conv_out = make_sparse_convmodule(
            encoder_out_channels, self.output_channels, kernel_size=(3, 1, 1), stride=(2, 1, 1),
            norm_cfg=norm_cfg, padding=0, indice_key='spconv_down2', conv_type='SparseConv3d')

    def forward(self, voxel_features, coors, batch_size):
        """voxel_features (N, C) f32, coors (N, 4) (batch, z, y, x), batch_size int or 0-d tensor
        -> (B, C*D, H, W) f32  (sparse_encoder.py:95-127)."""
        coors = coors.int()
        x = spconv.SparseConvTensor(voxel_features, coors, self.sparse_shape, batch_size)
        x = self.conv_input(x)
        if self.fp16_enabled:
            # the reference's auto_fp16 casts voxel_features to half before conv_input
            # (sparse_encoder.py:95); here the first layer stays fp32 (raw metric coordinates do
            # not survive 8 significant bits) and its OUTPUT is cast: every later layer runs on the
            # bf16 matrix cores, the last one writes fp32 for the dense map
            x = x.replace_feature(spconv.cast_bf16(x.features))
        encode_features = []
        for encoder_layer in self.encoder_layers:
            x = encoder_layer(x)
            encode_features.append(x)
        out = self.conv_out(encode_features[-1],
                            out_dtype=torch.float32 if self.fp16_enabled else None)
        spatial_features = out.dense()
        N, C, D, H, W = spatial_features.shape
        return spatial_features.view(N, C * D, H, W)

    def make_encoder_layers(self, make_block, norm_cfg, in_channels, block_type='conv_module',
                            conv_cfg=dict(type='SubMConv3d')):
        """sparse_encoder.py:129-203."""
        assert block_type in ['conv_module', 'basicblock']
        self.encoder_layers = spconv.SparseSequential()
        last_stage = len(self.encoder_channels) - 1
        for i, blocks in enumerate(self.encoder_channels):
            blocks = tuple(blocks)
            stage = []
            for j, out_channels in enumerate(blocks):
                padding = tuple(self.encoder_paddings[i])[j]
                down = dict(stride=2, padding=padding, indice_key=f'spconv{i + 1}',
                            conv_type='SparseConv3d')
                if block_type == 'conv_module':
                    if i != 0 and j == 0:   # every stage but the first opens with a strided conv
                        stage.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg,
                                                **down))
                    else:
                        stage.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg,
                                                padding=padding, indice_key=f'subm{i + 1}',
                                                conv_type='SubMConv3d'))
                elif j == len(blocks) - 1 and i != last_stage:  # basicblock stages END strided
                    stage.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg,
                                            **down))
                else:
                    stage.append(SparseBasicBlock(out_channels, out_channels, norm_cfg=norm_cfg,
                                                  conv_cfg=conv_cfg))
                in_channels = out_channels
            self.encoder_layers.add_module(f'encoder_layer{i + 1}', spconv.SparseSequential(*stage))
        return out_channels

"""`SparseUNet`: drop-in for mmdet3d/models/middle_encoders/sparse_unet.py:8-292 (Part-A2's
backbone; SURVEY 8f row 4, the last item: inverse convolutions on cached rulebooks).

Encoder = SparseEncoder's `conv_module` stages (same state-dict keys); decoder stage i =
`lateral_layer{i}` (SparseBasicBlock on the encoder tensor), channel concat with the tensor from
below, `merge_layer{i}` (SubM conv), channel-group sum (`reduce_channel`), `upsample_layer{i}`
(SparseInverseConv3d coupled to `spconv{i}`; a SubM conv in stage 1).  Every convolution runs on the
hand-written kernels; the concat / group sum / add between them are torch ops on the device, as in the
reference."""
import torch

from . import spconv
from .registry import MIDDLE_ENCODERS
from .sparse_block import SparseBasicBlock, make_sparse_convmodule
from .sparse_encoder import SparseEncoder


@MIDDLE_ENCODERS.register_module()
class SparseUNet(SparseEncoder):

    def __init__(self, in_channels, sparse_shape, order=('conv', 'norm', 'act'),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16,
                 output_channels=128,
                 encoder_channels=((16, ), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1, ), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)),
                 decoder_channels=((64, 64, 64), (64, 64, 32), (32, 32, 16), (16, 16, 16)),
                 decoder_paddings=((1, 0), (1, 0), (0, 0), (0, 1))):
        super().__init__(in_channels, sparse_shape, order=order, norm_cfg=norm_cfg,
                         base_channels=base_channels, output_channels=output_channels,
                         encoder_channels=encoder_channels, encoder_paddings=encoder_paddings,
                         block_type='conv_module')
        self.decoder_channels = decoder_channels
        self.decoder_paddings = decoder_paddings
        self.make_decoder_layers(make_sparse_convmodule, norm_cfg,
                                 tuple(self.encoder_channels[-1])[-1])

    def make_decoder_layers(self, make_block, norm_cfg, in_channels):
        """sparse_unet.py:237-292."""
        stages = len(self.decoder_channels)
        for i, (c_lateral, c_merge, c_up) in enumerate(self.decoder_channels):
            level = stages - i
            pad_merge, pad_up = self.decoder_paddings[i]
            setattr(self, f'lateral_layer{level}', SparseBasicBlock(
                in_channels, c_lateral, norm_cfg=norm_cfg,
                conv_cfg=dict(type='SubMConv3d', indice_key=f'subm{level}')))
            setattr(self, f'merge_layer{level}', make_block(
                in_channels * 2, c_merge, 3, norm_cfg=norm_cfg, padding=pad_merge,
                indice_key=f'subm{level}', conv_type='SubMConv3d'))
            if level != 1:
                up = make_block(in_channels, c_up, 3, norm_cfg=norm_cfg,
                                indice_key=f'spconv{level}', conv_type='SparseInverseConv3d')
            else:   # the finest level has no couple conv: a SubM conv instead
                up = make_block(in_channels, c_up, 3, norm_cfg=norm_cfg, padding=pad_up,
                                indice_key='subm1', conv_type='SubMConv3d')
            setattr(self, f'upsample_layer{level}', up)
            in_channels = c_up

    @staticmethod
    def reduce_channel(x, out_channels):
        """sparse_unet.py:171-189: sum groups of in/out consecutive channels."""
        n, c = x.features.shape
        assert c % out_channels == 0 and c >= out_channels
        x.features = x.features.view(n, out_channels, -1).sum(dim=2)
        return x

    def decoder_layer_forward(self, x_lateral, x_bottom, lateral_layer, merge_layer, upsample_layer):
        """sparse_unet.py:149-169."""
        x = lateral_layer(x_lateral)
        x.features = torch.cat((x_bottom.features, x.features), dim=1)
        x_merge = merge_layer(x)
        x = self.reduce_channel(x, x_merge.features.shape[1])
        x.features = x_merge.features + x.features
        return upsample_layer(x)

    def forward(self, voxel_features, coors, batch_size):
        """-> dict(spatial_features (B, C*D, H, W), seg_features (N, C_dec))  (sparse_unet.py:96-147)."""
        if self.fp16_enabled:
            raise NotImplementedError('SparseUNet runs fp32 (Part-A2 is outside the bf16 configs)')
        x = spconv.SparseConvTensor(voxel_features, coors.int(), self.sparse_shape, batch_size)
        x = self.conv_input(x)
        encode_features = []
        for encoder_layer in self.encoder_layers:
            x = encoder_layer(x)
            encode_features.append(x)
        dense = self.conv_out(encode_features[-1]).dense()
        N, C, D, H, W = dense.shape
        x = encode_features[-1]
        for i in range(self.stage_num, 0, -1):
            x = self.decoder_layer_forward(encode_features[i - 1], x,
                                           getattr(self, f'lateral_layer{i}'),
                                           getattr(self, f'merge_layer{i}'),
                                           getattr(self, f'upsample_layer{i}'))
        return dict(spatial_features=dense.view(N, C * D, H, W), seg_features=x.features)

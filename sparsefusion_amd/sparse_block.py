"""`SparseBasicBlock` and `make_sparse_convmodule`: drop-ins for mmdet3d/ops/sparse_block.py:81-204.

The reference block inherits mmdet's BasicBlock (conv1/bn1/conv2/bn2/relu attributes and
state-dict keys, tests/test_models/test_common_modules/test_sparse_unet.py:72-79); mmdet is not
available here, so the block is written out directly with the same attribute names.
"""
from torch import nn

from . import spconv
from .registry import build_conv_layer, build_norm_layer


def replace_feature(out, new_features):
    """sparse_block.py:7-13."""
    return out.replace_feature(new_features)


class SparseBasicBlock(spconv.SparseModule):
    """relu(bn2(conv2(relu(bn1(conv1(x))))) + x)  (sparse_block.py:116-139).

    In eval mode each half runs as one fused kernel (conv+BN+ReLU, conv+BN+residual+ReLU)."""

    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, conv_cfg=None, norm_cfg=None):
        super().__init__()
        if downsample is not None:
            raise NotImplementedError('downsample branches are not used by SparseEncoder')
        conv_cfg = conv_cfg or dict(type='SubMConv3d')
        norm_cfg = norm_cfg or dict(type='BN1d')
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 3, stride=stride, padding=1,
                                      dilation=1, bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, padding=1, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    @property
    def norm2(self):
        return getattr(self, self.norm2_name)

    def forward(self, x):
        identity = x.features
        assert x.features.dim() == 2, f'x.features.dim()={x.features.dim()}'
        if spconv._bn_foldable(self.norm1) and spconv._bn_foldable(self.norm2):
            s1, b1 = spconv.fold_bn(self.norm1)
            out = self.conv1.forward_fused(x, s1, b1, None, True,
                                           bn_params=(self.norm1.weight, self.norm1.bias))
            s2, b2 = spconv.fold_bn(self.norm2)
            return self.conv2.forward_fused(out, s2, b2, identity, True,
                                            bn_params=(self.norm2.weight, self.norm2.bias))
        out = self.conv1(x)
        out = replace_feature(out, self.relu(self.norm1(out.features)))
        out = self.conv2(out)
        out = replace_feature(out, self.norm2(out.features))
        out = replace_feature(out, self.relu(out.features + identity))
        return out


def make_sparse_convmodule(in_channels, out_channels, kernel_size, indice_key, stride=1, padding=0,
                           conv_type='SubMConv3d', norm_cfg=None, order=('conv', 'norm', 'act')):
    """sparse_block.py:142-204: SparseSequential of conv / norm / ReLU in `order`."""
    assert isinstance(order, tuple) and len(order) <= 3
    assert set(order) | {'conv', 'norm', 'act'} == {'conv', 'norm', 'act'}
    layers = []
    for layer in order:
        if layer == 'conv':
            if 'Inverse' in conv_type:      # sparse_block.py:191-198: no stride / padding argument
                layers.append(build_conv_layer(dict(type=conv_type, indice_key=indice_key),
                                               in_channels, out_channels, kernel_size, bias=False))
            else:
                layers.append(build_conv_layer(dict(type=conv_type, indice_key=indice_key),
                                               in_channels, out_channels, kernel_size,
                                               stride=stride, padding=padding, bias=False))
        elif layer == 'norm':
            layers.append(build_norm_layer(norm_cfg, out_channels)[1])
        elif layer == 'act':
            layers.append(nn.ReLU(inplace=True))
    return spconv.SparseSequential(*layers)

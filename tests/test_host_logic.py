"""Host-side logic that needs no GPU: module tree / state-dict keys, weight-layout handling,
registry entry points, BN folding, LidarBackbone planning."""
import pytest
import numpy as np
import torch

from helpers import encoder_cfg, torch_state_dict, weight_cfg
from sparsefusion_amd import registry, spconv, synth
from sparsefusion_amd.sparse_block import SparseBasicBlock, make_sparse_convmodule
from sparsefusion_amd.sparse_encoder import SparseEncoder
from sparsefusion_amd.voxelize import Voxelization


def test_state_dict_keys_match_reference_names():
    cfg = encoder_cfg('nusc')
    enc = SparseEncoder(**cfg)
    keys = set(enc.state_dict().keys())
    sd = synth.encoder_state_dict(0, 5, **weight_cfg(cfg))
    assert keys == set(sd.keys())
    assert 'encoder_layers.encoder_layer3.2.0.weight' in keys
    assert 'encoder_layers.encoder_layer4.1.bn2.running_var' in keys
    assert enc.state_dict()['conv_out.0.weight'].shape == (128, 3, 1, 1, 128)   # spconv-2 layout
    enc.load_state_dict(torch_state_dict(sd))
    conv = enc.encoder_layers.encoder_layer3[2][0]
    assert isinstance(conv, spconv.SparseConv3d) and conv.padding == [0, 1, 1] and conv.stride == [2, 2, 2]
    assert conv.indice_key == 'spconv3' and enc.conv_input[0].indice_key == 'subm1'
    blk = enc.encoder_layers.encoder_layer1[0]
    assert isinstance(blk, SparseBasicBlock) and blk.conv1.indice_key is None
    assert blk.bn1.eps == 1e-3 and blk.bn1.momentum == 0.01


def test_spconv1_weight_layout_is_permuted_on_load():
    conv = spconv.SubMConv3d(4, 8, 3, bias=False)
    w1 = torch.randn(3, 3, 3, 4, 8)                       # spconv-1 (kD,kH,kW,Cin,Cout)
    conv.load_state_dict({'weight': w1})
    assert torch.equal(conv.weight.data, w1.permute(4, 0, 1, 2, 3))
    w2 = torch.randn(8, 3, 3, 3, 4)
    conv.load_state_dict({'weight': w2})
    assert torch.equal(conv.weight.data, w2)


def test_conv_module_variant_and_registry():
    enc = registry.build_middle_encoder(dict(type='SparseEncoder', in_channels=4,
                                             sparse_shape=[41, 1600, 1408]))   # class defaults
    stage2 = enc.encoder_layers.encoder_layer2
    assert isinstance(stage2[0][0], spconv.SparseConv3d) and stage2[0][0].indice_key == 'spconv2'
    assert isinstance(stage2[1][0], spconv.SubMConv3d) and stage2[1][0].indice_key == 'subm2'
    assert enc.encoder_layers.encoder_layer4[0][0].padding == [0, 1, 1]
    vfe = registry.build_voxel_encoder(dict(type='HardSimpleVFE', num_features=5))
    assert vfe.num_features == 5
    m = make_sparse_convmodule(4, 8, 3, 'k', conv_type='SubMConv3d',
                               norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01))
    assert [type(x).__name__ for x in m] == ['SubMConv3d', 'BatchNorm1d', 'ReLU']
    assert m[0].bias is None


def test_fold_bn_equals_batchnorm_eval():
    bn = torch.nn.BatchNorm1d(6, eps=1e-3).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
        x = torch.randn(10, 6)
        scale, shift = spconv.fold_bn(bn)
        assert torch.allclose(x * scale + shift, bn(x), rtol=1e-5, atol=1e-6)


def test_voxelization_attributes():
    v = Voxelization([0.075, 0.075, 0.2], [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], 10, (120000, 160000))
    assert v.grid_size.tolist() == [1440, 1440, 40] and v.max_voxels == (120000, 160000)
    assert [int(x) for x in v.pcd_shape] == [1, 1440, 1440]
    assert Voxelization([1, 1, 1], [0, 0, 0, 2, 2, 2], 5, 7).max_voxels == (7, 7)
    assert 'max_num_points=10' in repr(v)


def test_conv_output_size_rule():
    assert spconv.get_conv_output_size([41, 1440, 1440], [3] * 3, [2] * 3, [1] * 3, [1] * 3) == [21, 720, 720]
    assert spconv.get_conv_output_size([11, 360, 360], [3] * 3, [2] * 3, [0, 1, 1], [1] * 3) == [5, 180, 180]
    assert spconv.get_conv_output_size([5, 180, 180], [3, 1, 1], [2, 1, 1], [0] * 3, [1] * 3) == [2, 180, 180]


def test_ordered_kernel_selection(monkeypatch):
    """spconv.ORDERED_SHAPES: which fp32 layer shapes run on the mask-ordered kernel (default: the
    64- and 128-channel SubM shapes, where it measured faster end to end)."""
    assert spconv.use_ordered(64, 64) and spconv.use_ordered(128, 128)
    assert not spconv.use_ordered(32, 32) and not spconv.use_ordered(64, 128)
    assert not spconv.use_ordered(128, 128, subm=False)      # a strided rulebook is used once
    monkeypatch.setattr(spconv, 'ORDERED_SHAPES', '32x32,64x64')
    assert spconv.use_ordered(64, 64) and spconv.use_ordered(32, 32)
    assert not spconv.use_ordered(128, 128) and not spconv.use_ordered(8, 16)
    assert not spconv.use_ordered(64, 64, subm=False)
    monkeypatch.setattr(spconv, 'ORDERED_SHAPES', 'all')
    assert spconv.use_ordered(128, 64) and not spconv.use_ordered(8, 16)   # first layer: cin < 16
    assert spconv.use_ordered(64, 128, subm=False)
    monkeypatch.setattr(spconv, 'ORDERED_SHAPES', 'none')
    assert not spconv.use_ordered(64, 64)


def test_registry_builders_resolve_every_mirrored_module():
    for kind, types in (('middle', ['SparseEncoder', 'SparseUNet', 'PointPillarsScatter']),
                        ('voxel', ['HardSimpleVFE', 'HardVFE', 'PillarFeatureNet'])):
        reg = registry.MIDDLE_ENCODERS if kind == 'middle' else registry.VOXEL_ENCODERS
        registry._populate()
        for t in types:
            assert reg.get(t) is not None, t
    assert registry.build_backbone(dict(type='SECOND', in_channels=8, out_channels=[8],
                                        layer_nums=[1], layer_strides=[1])) is not None
    assert registry.build_neck(dict(type='SECONDFPN', in_channels=[8], out_channels=[8],
                                    upsample_strides=[1])) is not None


def test_forward_only_guard_raises_in_backward():
    """Ops behind the C ABI are forward-only: an output computed from a tensor that requires grad
    must carry a grad_fn whose backward raises (instead of silently skipping the weights)."""
    import pytest
    from sparsefusion_amd import _lib
    w = torch.nn.Parameter(torch.ones(3))
    out = torch.zeros(3)                       # stands for a buffer the C ABI wrote
    y = _lib.forward_only(out, None, w)
    assert y.requires_grad and y.data_ptr() == out.data_ptr()
    with pytest.raises(NotImplementedError, match='forward-only'):
        y.sum().backward()
    with torch.no_grad():
        assert _lib.forward_only(out, w) is out
    w.requires_grad_(False)                    # frozen LiDAR branch: nothing to guard
    assert _lib.forward_only(out, w) is out


def test_vfe_decorated_channels():
    from sparsefusion_amd.voxel_encoder import decorated_channels
    assert decorated_channels(5, 8) == 5                   # mask only
    assert decorated_channels(5, 1 | 2 | 8) == 11          # HardVFE: cluster + voxel centre
    assert decorated_channels(4, 1 | 2 | 16 | 32 | 8) == 9  # PillarFeatureNet: +3 +2
    assert decorated_channels(4, 1 | 2 | 4 | 16 | 8) == 10  # ... + distance


def test_sparse_sequential_container():
    """The container contract of modules.py:55-123: positional / ordered-dict / keyword layers,
    integer indexing (negative from the end), add(), duplicate names rejected."""
    import torch.nn as nn
    from collections import OrderedDict
    seq = spconv.SparseSequential(nn.ReLU(), nn.Identity(), tail=nn.ReLU())
    assert len(seq) == 3 and isinstance(seq[1], nn.Identity) and seq[-1] is seq[2]
    assert list(seq._modules) == ['0', '1', 'tail']
    seq.add(nn.Identity())
    seq.add(nn.ReLU(), 'named')
    assert list(seq._modules) == ['0', '1', 'tail', '3', 'named'] and len(seq) == 5
    od = spconv.SparseSequential(OrderedDict(a=nn.ReLU(), b=nn.Identity()))
    assert isinstance(od[0], nn.ReLU) and isinstance(od[-1], nn.Identity)
    with pytest.raises(IndexError):
        seq[5]
    with pytest.raises(IndexError):
        seq[-6]
    with pytest.raises(ValueError):
        spconv.SparseSequential(nn.ReLU(), **{'0': nn.ReLU()})
    with pytest.raises(ValueError):
        seq.add(nn.ReLU(), 'tail')

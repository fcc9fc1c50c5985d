"""SECOND / SECONDFPN on the CPU: the oracle's transcription of the reference's
torch module chains (oracle.second_forward / second_fpn_forward) is pinned against a direct numpy
statement of cross-correlation and of the kernel = stride transposed convolution, and the
modules' constructors are checked on the reference's own construction tests
(tests/test_models/test_necks/test_fpn.py:4-37, test_necks.py:7-45).  No kernel runs here."""
import numpy as np
import pytest
import torch

from sparsefusion_amd import synth
from sparsefusion_amd.second import BACKBONES, NECKS, SECOND, SECONDFPN


def direct_conv2d(x, w, stride, pad):
    B, C, H, W = x.shape
    co, ci, k, _ = w.shape
    xp = np.zeros((B, C, H + 2 * pad, W + 2 * pad), np.float64)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    ho, wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = np.zeros((B, co, ho, wo), np.float64)
    for ky in range(k):
        for kx in range(k):
            patch = xp[:, :, ky:ky + stride * ho:stride, kx:kx + stride * wo:stride]
            y += np.einsum('bchw,oc->bohw', patch, w[:, :, ky, kx].astype(np.float64))
    return y


def direct_deconv2d(x, w, s):
    B, C, H, W = x.shape
    ci, co, k, _ = w.shape
    assert k == s
    y = np.zeros((B, co, H * s, W * s), np.float64)
    for dy in range(s):
        for dx in range(s):
            y[:, :, dy::s, dx::s] = np.einsum('bchw,co->bohw', x.astype(np.float64),
                                              w[:, :, dy, dx].astype(np.float64))
    return y


def bn_relu(y, bn, eps):
    sh = (1, -1, 1, 1)
    y = (y - bn['running_mean'].reshape(sh)) / np.sqrt(bn['running_var'].reshape(sh) + eps)
    return np.maximum(y * bn['weight'].reshape(sh) + bn['bias'].reshape(sh), 0)


@pytest.mark.parametrize('stride,pad,k', [(1, 1, 3), (2, 1, 3), (1, 0, 1), (2, 0, 2)])
def test_conv_transcription(oracle, stride, pad, k):
    rng = np.random.default_rng(k * 10 + stride)
    x = rng.normal(0, 1, (2, 5, 11, 14)).astype(np.float32)
    w = rng.normal(0, 0.3, (7, 5, k, k)).astype(np.float32)
    bn = dict(weight=rng.uniform(0.5, 1.5, 7), bias=rng.normal(0, 0.1, 7),
              running_mean=rng.normal(0, 0.1, 7), running_var=rng.uniform(0.5, 1.5, 7))
    got = oracle.conv2d_bn_relu(x, w, bn, stride, pad, 1e-3)
    np.testing.assert_allclose(got, bn_relu(direct_conv2d(x, w, stride, pad), bn, 1e-3),
                               rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('s', [1, 2, 4])
def test_deconv_transcription(oracle, s):
    rng = np.random.default_rng(s)
    x = rng.normal(0, 1, (2, 6, 7, 9)).astype(np.float32)
    w = rng.normal(0, 0.3, (6, 4, s, s)).astype(np.float32)
    bn = dict(weight=rng.uniform(0.5, 1.5, 4), bias=rng.normal(0, 0.1, 4),
              running_mean=rng.normal(0, 0.1, 4), running_var=rng.uniform(0.5, 1.5, 4))
    got = oracle.conv2d_bn_relu(x, w, bn, s, 0, 1e-3, transposed=True)
    np.testing.assert_allclose(got, bn_relu(direct_deconv2d(x, w, s), bn, 1e-3), rtol=1e-4,
                               atol=1e-5)


def test_secfpn_construction():
    """tests/test_models/test_necks/test_fpn.py:4-37."""
    neck = NECKS.build(dict(type='SECONDFPN', in_channels=[2, 3], upsample_strides=[1, 2],
                            out_channels=[4, 6]))
    assert neck.deblocks[0][0].in_channels == 2
    assert neck.deblocks[1][0].in_channels == 3
    assert neck.deblocks[0][0].out_channels == 4
    assert neck.deblocks[1][0].out_channels == 6
    assert neck.deblocks[0][0].stride == (1, 1)
    assert neck.deblocks[1][0].stride == (2, 2)
    with pytest.raises(AssertionError):
        NECKS.build(dict(type='SECONDFPN', in_channels=[2, 2], upsample_strides=[1, 2, 4],
                         out_channels=[2, 2]))
    with pytest.raises(AssertionError):
        NECKS.build(dict(type='SECONDFPN', in_channels=[2, 2, 4], upsample_strides=[1, 2, 4],
                         out_channels=[2, 2]))


def test_state_dict_keys_match_reference_layout():
    """second.py:36-62 / second_fpn.py:42-65: nn.Sequential indices conv 3j, BN 3j+1."""
    from sparsefusion_amd import configs
    bcfg = dict(configs.NUSC['pts_backbone'])
    ncfg = dict(configs.NUSC['pts_neck'])
    second, neck = BACKBONES.build(bcfg), NECKS.build(ncfg)
    bcfg.pop('type'), ncfg.pop('type')
    sd = synth.second_state_dict(1, **bcfg)
    assert set(sd) == set(second.state_dict())
    for k, v in second.state_dict().items():
        assert tuple(v.shape) == tuple(np.shape(sd[k])), k
    nsd = synth.second_fpn_state_dict(2, **ncfg)
    assert set(nsd) == set(neck.state_dict())
    for k, v in neck.state_dict().items():
        assert tuple(v.shape) == tuple(np.shape(nsd[k])), k
    assert isinstance(neck.deblocks[0][0], torch.nn.Conv2d)            # use_conv_for_no_stride
    assert isinstance(neck.deblocks[1][0], torch.nn.ConvTranspose2d)
    assert second.blocks[1][0].stride == (2, 2) and second.blocks[0][0].stride == (1, 1)


def test_no_cpu_path():
    second = SECOND(in_channels=8, out_channels=[8], layer_nums=[1], layer_strides=[1])
    with pytest.raises(RuntimeError, match='no CPU path|HIP device'):
        second.eval()(torch.zeros(1, 8, 4, 4))
    assert isinstance(SECONDFPN(in_channels=[8], out_channels=[8], upsample_strides=[1]), SECONDFPN)

"""Sparse convolution / SparseEncoder features on the GPU vs the CPU oracle (fp32).

Tolerance: north_star asks rtol 1e-4 in fp32.  The kernel and the oracle sum the same terms in
different orders (MFMA fp32 FMA chains vs plain C loops), so results are compared elementwise
with |a - b| <= 1e-4*|b| + 1e-5*max|b| (helpers.assert_close), and structure bit for bit."""
import numpy as np
import pytest
import torch

from helpers import (SMALL_RANGE, SMALL_SHAPE, SMALL_VOXEL, assert_close, encoder_cfg, rel_err,
                     small_points,
                     to_dev, torch_state_dict, weight_cfg)
from sparsefusion_amd import spconv, synth
from sparsefusion_amd.sparse_encoder import SparseEncoder
from test_gpu_rulebook import active_set

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.mark.parametrize('cin,cout', [(5, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64),
                                      (64, 128), (128, 128), (64, 16), (3, 32)])
@pytest.mark.parametrize('subm', [True, False])
def test_indice_conv_matches_oracle(dev, oracle, cin, cout, subm):
    rng = np.random.default_rng(cin * 131 + cout)
    idx = active_set(oracle, 5, 1500, 2)
    ks, st, pd = (3, 1, 1) if subm else (3, 2, 1)
    outids, pairs, num, _ = oracle.get_indice_pairs(idx, 2, SMALL_SHAPE, ks, st, pd, 1, subm)
    feats = rng.normal(0, 1, (idx.shape[0], cin)).astype(np.float32)
    w = rng.normal(0, 0.2, (3, 3, 3, cin, cout)).astype(np.float32)
    expect = oracle.indice_conv(feats, w.reshape(27, cin, cout), pairs, num, outids.shape[0], subm)
    got = spconv.indice_conv(to_dev(feats, dev), to_dev(w, dev), to_dev(pairs, dev),
                             to_dev(num, dev), outids.shape[0], False, subm)
    torch.cuda.synchronize()
    assert_close(got.cpu().numpy(), expect, RTOL)


def test_fused_epilogue_matches_oracle(dev, oracle):
    rng = np.random.default_rng(9)
    idx = active_set(oracle, 6, 2000, 1)
    outids, pairs, num, _ = oracle.get_indice_pairs(idx, 1, SMALL_SHAPE, 3, 1, 1, 1, True)
    n, c = idx.shape[0], 32
    feats = rng.normal(0, 1, (n, c)).astype(np.float32)
    w = rng.normal(0, 0.1, (27, c, c)).astype(np.float32)
    bn = dict(weight=rng.uniform(0.5, 1.5, c), bias=rng.normal(0, 0.1, c),
              running_mean=rng.normal(0, 0.1, c), running_var=rng.uniform(0.5, 1.5, c))
    bn = {k: v.astype(np.float32) for k, v in bn.items()}
    y = oracle.indice_conv(feats, w, pairs, num, n, True)
    expect = oracle.bn_act(y, bn, 1e-3, residual=feats, relu=True)
    x = spconv.SparseConvTensor(to_dev(feats, dev), to_dev(idx, dev), SMALL_SHAPE, 1)
    rb = spconv.build_rulebook(x.coords(), 3, 1, 1, 1, True)
    scale = to_dev(bn['weight'] / np.sqrt(bn['running_var'] + np.float32(1e-3)), dev)
    shift = to_dev(bn['bias'], dev) - to_dev(bn['running_mean'], dev) * scale
    got = spconv.sparse_conv(x.features, spconv.pack_filters(to_dev(w, dev)), c, c, rb, scale,
                             shift.contiguous(), x.features, True)
    torch.cuda.synchronize()
    assert_close(got.cpu().numpy(), expect, RTOL)


def make_encoder(dev, in_channels, seed=0, name='nusc', **over):
    cfg = encoder_cfg(name)
    cfg.update(over)
    cfg['in_channels'] = in_channels
    cfg['sparse_shape'] = SMALL_SHAPE
    sd = synth.encoder_state_dict(seed, in_channels, **weight_cfg(cfg))
    enc = SparseEncoder(**cfg)
    enc.load_state_dict(torch_state_dict(sd))
    return enc.to(dev).eval(), sd, cfg


@pytest.mark.parametrize('batch', [1, 2])
@pytest.mark.parametrize('block_type', ['basicblock', 'conv_module'])
def test_sparse_encoder_matches_oracle(dev, oracle, batch, block_type):
    over = dict(block_type=block_type)
    if block_type == 'conv_module':
        over.update(encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                    encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)))
    enc, sd, cfg = make_encoder(dev, 5, 3, **over)
    vox, coors = [], []
    for b in range(batch):
        pts = small_points(40 + b, 3000 + 500 * b)
        v, c, n = oracle.hard_voxelize(pts, SMALL_VOXEL, SMALL_RANGE, 10, 20000)
        vox.append(oracle.vfe_mean(v, n, 5))
        coors.append(c)
    feats = np.concatenate(vox, 0)
    coors4 = oracle.batch_coors(coors)
    trace = {}
    expect = oracle.sparse_encoder_forward(
        sd, feats, coors4, batch, SMALL_SHAPE, 5, bn_eps=1e-3, trace=trace, **weight_cfg(cfg),
        encoder_paddings=cfg['encoder_paddings'])
    with torch.no_grad():
        got = enc(to_dev(feats, dev), to_dev(coors4, dev), torch.tensor(batch, device=dev))
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert got.shape == expect.shape == (batch, 256, SMALL_SHAPE[1] // 8, SMALL_SHAPE[2] // 8)
    assert_close(got, expect, RTOL)
    assert rel_err(got, expect) <= RTOL
    assert np.abs(expect).max() > 1e-3   # the comparison is not vacuous


def test_encoder_train_mode_bn_unfused_path(dev, oracle):
    """Training-mode BatchNorm (batch statistics) takes the unfused module path; check it against
    the same modules evaluated piecewise in torch on the GPU conv outputs."""
    enc, sd, cfg = make_encoder(dev, 5, 5)
    pts = small_points(77, 2500)
    v, c, n = oracle.hard_voxelize(pts, SMALL_VOXEL, SMALL_RANGE, 10, 20000)
    feats, coors4 = oracle.vfe_mean(v, n, 5), oracle.batch_coors([c])
    x = spconv.SparseConvTensor(to_dev(feats, dev), to_dev(coors4, dev), SMALL_SHAPE, 1)
    mod = enc.conv_input
    mod.train()
    with torch.no_grad():
        y = mod(x).features
        conv = mod[0](x).features
        bn = torch.nn.functional.batch_norm(conv, None, None, mod[1].weight, mod[1].bias, True, 0.0,
                                            mod[1].eps)
        ref = torch.relu(bn)
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-6)

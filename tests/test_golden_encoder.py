"""Committed golden vectors for the SparseEncoder path (tests/golden/encoder_b{1,2}.npz, made by
tests/golden/make_golden.py with the REFERENCE's compiled spconv-1.x CPU code -- oracle/_ref/spconv:
geometry.h rulebooks, reordering.cc gather / scatter-add + torch::mm_out -- inside the oracle's
layer wiring): the oracle must reproduce them (CPU: rulebooks bit for bit, BEV to 1e-5, the GEMM
summation order being the only difference), and the HIP path must match them (GPU): rulebook
counts and coordinates bit for bit, BEV within 1e-4.  B=2 has unequal sample sizes (batch-index handling, detectors/sparsefusion.py:153-157)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from helpers import SMALL_SHAPE, assert_close, to_dev, torch_state_dict
from sparsefusion_amd import spconv, synth
from sparsefusion_amd.sparse_encoder import SparseEncoder

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
CFG = dict(encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
           encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)), block_type='basicblock')


def load(name):
    g = np.load(os.path.join(GOLD, name + '.npz'))
    bev = np.zeros(int(np.prod(g['bev_shape'])), np.float32)
    bev[g['bev_nonzero_idx']] = g['bev_nonzero_val']
    return g, bev.reshape(tuple(g['bev_shape']))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize('name,batch', [('encoder_b1', 1), ('encoder_b2', 2)])
def test_oracle_reproduces_golden(oracle, name, batch):
    g, bev = load(name)
    sd = synth.encoder_state_dict(int(g['weight_seed']), 5, encoder_channels=CFG['encoder_channels'],
                                  block_type='basicblock')
    trace = {}
    got = oracle.sparse_encoder_forward(sd, g['voxel_features'], g['coors'], batch, SMALL_SHAPE, 5,
                                        trace=trace, **CFG)
    assert np.array_equal(got != 0, bev != 0)
    assert_close(got, bev, rtol=1e-5, atol_scale=1e-6)
    for i, k in enumerate(g['layer_names']):
        t = trace[str(k)]
        assert t['outids'].shape[0] == g['layer_rows'][i]
        assert np.array_equal(np.pad(t['num'], (0, 27 - len(t['num']))), g['layer_indice_num'][i])
        assert sha(t['outids']) == str(g['layer_outids_sha'][i])
        assert sha(t['pairs']) == str(g['layer_pairs_sha'][i])


@pytest.mark.gpu
@pytest.mark.parametrize('name,batch', [('encoder_b1', 1), ('encoder_b2', 2)])
def test_hip_path_matches_golden(dev, name, batch):
    g, bev = load(name)
    sd = synth.encoder_state_dict(int(g['weight_seed']), 5, encoder_channels=CFG['encoder_channels'],
                                  block_type='basicblock')
    enc = SparseEncoder(in_channels=5, sparse_shape=SMALL_SHAPE, **CFG)
    enc.load_state_dict(torch_state_dict(sd))
    enc = enc.to(dev).eval()
    feats, coors = to_dev(g['voxel_features'], dev), to_dev(g['coors'], dev)
    with torch.no_grad():
        got = enc(feats, coors, batch)
        # walk the layers once more to compare every rulebook with the golden hashes
        x = spconv.SparseConvTensor(feats, coors, SMALL_SHAPE, batch)
        mods = [enc.conv_input] + [b for st in enc.encoder_layers for b in st] + [enc.conv_out]
        for i, m in enumerate(mods):
            conv = m.conv1 if hasattr(m, 'conv1') else m[0]
            rb = conv.rulebook_for(x)
            pairs, num = rb.indice_pairs()
            assert rb.out_coords.n == g['layer_rows'][i]
            assert np.array_equal(np.pad(num.cpu().numpy(), (0, 27 - num.numel())),
                                  g['layer_indice_num'][i])
            assert sha(rb.out_coords.indices.cpu().numpy()) == str(g['layer_outids_sha'][i])
            assert sha(pairs.cpu().numpy()) == str(g['layer_pairs_sha'][i])
            x = m(x)
    assert_close(got.cpu().numpy(), bev, 1e-4)

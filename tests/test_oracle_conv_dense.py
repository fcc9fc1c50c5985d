"""CPU oracle pinning, sparse-conv part (runs without a GPU).

The reference's own tests hold no value fixtures for SparseEncoder (shapes only,
tests/test_models/test_common_modules/test_middle_encoders.py:25-26) and its spconv 1.x sources
cannot be built here (they need <cuda_runtime_api.h>), so the rulebook + convolution restatement
is pinned two ways:
  * pair examples recorded from the compiled reference during the survey:
    SubM k=1: in (12,819,131) -> out (13,820,131); strided k=3: in (13,820,131) -> out (7,410,66);
  * dense equivalence: rulebook convention == plain cross-correlation, so the sparse result must
    equal torch.nn.functional.conv3d on the densified input at the active outputs."""
import numpy as np
import torch
import torch.nn.functional as F

from helpers import small_points, SMALL_RANGE, SMALL_VOXEL


def test_recorded_reference_pairs(oracle):
    idx = np.asarray([[0, 12, 819, 131], [0, 13, 820, 131]], np.int32)
    shape = [41, 1440, 1440]
    out, pairs, num, _ = oracle.get_indice_pairs(idx, 1, shape, 3, 1, 1, 1, True)
    # offset k=1 = (kz,ky,kx) = (0,0,1): in = out - 1 + (0,0,1)
    assert num[1] == 1 and pairs[1, 0, 0] == 0 and pairs[1, 1, 0] == 1
    assert num[13] == 2 and num.sum() == 4
    out, pairs, num, oshape = oracle.get_indice_pairs(idx[1:], 1, shape, 3, 2, 1, 1, False)
    assert oshape == [21, 720, 720]
    k3 = [tuple(out[pairs[3, 1, i]]) for i in range(num[3])]
    assert (0, 7, 410, 66) in k3


def _dense_check(oracle, ks, st, pd, subm, cin, cout, batch=2, seed=0):
    rng = np.random.default_rng(seed)
    shape = [12, 20, 24]
    n = 300
    coords = np.unique(np.stack([rng.integers(0, batch, n), rng.integers(0, shape[0], n),
                                 rng.integers(0, shape[1], n), rng.integers(0, shape[2], n)], 1),
                       axis=0).astype(np.int32)
    coords = coords[rng.permutation(coords.shape[0])]
    feats = rng.normal(0, 1, (coords.shape[0], cin)).astype(np.float32)
    tri = lambda v: list(v) if isinstance(v, (tuple, list)) else [v] * 3
    ks3, st3, pd3 = tri(ks), tri(st), tri(pd)
    w = rng.normal(0, 0.3, (*ks3, cin, cout)).astype(np.float32)          # spconv-1 layout
    out, pairs, num, oshape = oracle.get_indice_pairs(coords, batch, shape, ks, st, pd, 1, subm)
    K = int(np.prod(ks3))
    y = oracle.indice_conv(feats, w.reshape(K, cin, cout), pairs, num, out.shape[0], subm)
    dense = np.zeros((batch, cin, *shape), np.float32)
    dense[coords[:, 0], :, coords[:, 1], coords[:, 2], coords[:, 3]] = feats
    wt = torch.from_numpy(np.transpose(w, (4, 3, 0, 1, 2)).copy())       # (co,ci,kz,ky,kx)
    pad = [k // 2 for k in ks3] if subm else pd3
    ref = F.conv3d(torch.from_numpy(dense).double(), wt.double(), stride=(1, 1, 1) if subm else
                   tuple(st3), padding=tuple(pad)).numpy()
    assert list(ref.shape[2:]) == list(oshape)
    got = ref[out[:, 0], :, out[:, 1], out[:, 2], out[:, 3]]
    assert np.allclose(y, got, rtol=1e-4, atol=1e-4)
    if not subm:   # active output set == support reachable from the inputs
        reach = F.conv3d(torch.from_numpy((np.abs(dense).sum(1, keepdims=True) > 0)
                                          .astype(np.float64)),
                         torch.ones(1, 1, *ks3, dtype=torch.float64), stride=tuple(st3),
                         padding=tuple(pd3)).numpy()[:, 0] > 0
        mask = np.zeros_like(reach)
        mask[out[:, 0], out[:, 1], out[:, 2], out[:, 3]] = True
        assert np.array_equal(mask, reach)
        assert np.unique(out, axis=0).shape[0] == out.shape[0]


def test_subm_equals_dense_conv(oracle):
    _dense_check(oracle, 3, 1, 1, True, 5, 16)
    _dense_check(oracle, (1, 3, 3), 1, 0, True, 4, 8, seed=1)


def test_strided_equals_dense_conv(oracle):
    _dense_check(oracle, 3, 2, 1, False, 16, 32)
    _dense_check(oracle, 3, 2, (0, 1, 1), False, 8, 8, seed=2)
    _dense_check(oracle, (3, 1, 1), (2, 1, 1), 0, False, 8, 16, seed=3)
    _dense_check(oracle, 3, 1, 0, False, 4, 4, seed=4)


def test_bn_and_dense_statements(oracle):
    rng = np.random.default_rng(0)
    x = rng.normal(0, 1, (50, 16)).astype(np.float32)
    bn = dict(weight=rng.uniform(0.5, 1.5, 16), bias=rng.normal(0, 0.1, 16),
              running_mean=rng.normal(0, 0.1, 16), running_var=rng.uniform(0.5, 1.5, 16))
    bn = {k: v.astype(np.float32) for k, v in bn.items()}
    m = torch.nn.BatchNorm1d(16, eps=1e-3).eval()
    m.load_state_dict({**{k: torch.from_numpy(v) for k, v in bn.items()},
                       'num_batches_tracked': torch.tensor(0)})
    res = rng.normal(0, 1, x.shape).astype(np.float32)
    with torch.no_grad():
        ref = torch.relu(m(torch.from_numpy(x)) + torch.from_numpy(res)).numpy()
    assert np.allclose(oracle.bn_act(x, bn, 1e-3, res, True), ref, rtol=1e-5, atol=1e-6)
    idx = np.asarray([[0, 1, 2, 3], [1, 0, 4, 5]], np.int32)
    f = rng.normal(0, 1, (2, 3)).astype(np.float32)
    d = oracle.dense_bev(f, idx, 2, [2, 6, 7])
    assert d.shape == (2, 6, 6, 7)
    assert d[0, 0 * 2 + 1, 2, 3] == f[0, 0] and d[1, 2 * 2 + 0, 4, 5] == f[1, 2]
    assert np.count_nonzero(d) == 6


def test_encoder_wiring_matches_reference_module_tree(oracle):
    """State-dict prefixes / layer order of the oracle wiring for the nuScenes config
    (SURVEY 3.2 table)."""
    from sparsefusion_amd import configs
    cfg = dict(configs.NUSC['pts_middle_encoder'])
    specs = oracle.encoder_layer_specs(5, encoder_channels=cfg['encoder_channels'],
                                       encoder_paddings=cfg['encoder_paddings'],
                                       block_type='basicblock')
    names = [(s['prefix'], s['kind'], s['cin'], s['cout'], s['subm']) for s in specs]
    assert names[0] == ('conv_input', 'convmodule', 5, 16, True)
    assert names[3] == ('encoder_layers.encoder_layer1.2', 'convmodule', 16, 32, False)
    assert specs[9]['padding'] == [0, 1, 1] and specs[9]['cout'] == 128
    assert names[-1] == ('conv_out', 'convmodule', 128, 128, False)
    assert sum(1 for s in specs if s['kind'] == 'basicblock') == 8 and len(specs) == 13

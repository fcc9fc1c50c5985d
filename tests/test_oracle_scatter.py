"""The oracle's DynamicScatter restatement against the check the reference's own test makes
(tests/test_models/test_voxel_encoder/test_dynamic_scatter.py:12-51): output coordinates equal the
sorted unique non-negative input coordinates, features equal the per-voxel mean / max (its
tolerances: atol 1e-2, rtol 1e-5).  That test needs no fixture file -- it IS the pin."""
import numpy as np
import pytest


def make_case(seed, n, c, hi):
    rng = np.random.default_rng(seed)
    feats = (rng.random((n, c), dtype=np.float32) * 100 - 50).astype(np.float32)
    coors = rng.integers(-1, hi, size=(n, 3)).astype(np.int32)
    coors[coors.min(axis=-1) < 0] = -1          # test_dynamic_scatter.py:16
    return feats, coors


def numpy_reference(feats, coors):
    valid = coors.min(axis=-1) >= 0
    uc, inv = np.unique(coors[valid], axis=0, return_inverse=True)   # sorted, as coors.unique(dim=0)
    inv = inv.reshape(-1)
    f = feats[valid].astype(np.float64)
    s = np.zeros((uc.shape[0], feats.shape[1]))
    np.add.at(s, inv, f)
    cnt = np.bincount(inv, minlength=uc.shape[0])
    mx = np.full((uc.shape[0], feats.shape[1]), -np.inf)
    np.maximum.at(mx, inv, f)
    p2v = np.full(coors.shape[0], -1, np.int64)
    p2v[valid] = inv
    return uc, s, cnt, mx, p2v


@pytest.mark.parametrize('seed,n,c,hi', [(0, 20000, 3, 20), (1, 5000, 4, 3), (2, 300, 5, 40)])
def test_oracle_matches_reference_test_recipe(oracle, seed, n, c, hi):
    feats, coors = make_case(seed, n, c, hi)
    uc, s, cnt, mx, p2v = numpy_reference(feats, coors)
    vf, vc, m, k = oracle.dynamic_scatter(feats, coors, 'mean')
    assert np.array_equal(vc, uc)
    assert np.array_equal(m, p2v)
    assert np.array_equal(k, cnt)
    assert np.allclose(vf, s / cnt[:, None], atol=1e-2, rtol=1e-5)
    assert np.allclose(vf, s / cnt[:, None], atol=2e-4, rtol=1e-5)   # and much tighter than that
    vf, vc, m, k = oracle.dynamic_scatter(feats, coors, 'max')
    assert np.array_equal(vc, uc)
    assert np.array_equal(vf, mx.astype(np.float32))
    assert not k.any()                                                # count only filled for mean
    vf, vc, m, k = oracle.dynamic_scatter(feats, coors, 'sum')
    assert np.allclose(vf, s, atol=1e-2, rtol=1e-5)


def test_oracle_edge_cases(oracle):
    feats = np.ones((7, 2), np.float32)
    coors = np.full((7, 3), -1, np.int32)
    vf, vc, m, k = oracle.dynamic_scatter(feats, coors, 'mean')        # everything dropped
    assert vf.shape == (0, 2) and vc.shape == (0, 3) and (m == -1).all()
    vf, vc, m, k = oracle.dynamic_scatter(np.zeros((0, 2), np.float32), np.zeros((0, 3), np.int32), 'max')
    assert vf.shape[0] == 0 and m.shape[0] == 0
    coors = np.array([[0, 0, 5], [0, 0, 5], [0, 1, 0], [-1, 3, 3]], np.int32)   # one negative entry
    feats = np.array([[1, 2], [3, -4], [5, 6], [9, 9]], np.float32)
    vf, vc, m, k = oracle.dynamic_scatter(feats, coors, 'mean')
    assert vc.tolist() == [[0, 0, 5], [0, 1, 0]] and m.tolist() == [0, 0, 1, -1]
    assert vf.tolist() == [[2, -1], [5, 6]] and k.tolist() == [2, 1]

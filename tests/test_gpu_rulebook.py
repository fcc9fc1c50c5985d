"""Rulebooks built on the GPU (hash table + neighbour table) vs the CPU oracle's restatement of
spconv 1.x getIndicePairsSubM / getIndicePairsConv: bit-exact, including the CPU reference's
row ORDER (outputs in first-touch order, pairs in input order), and in canonical form
(SURVEY 8a-R)."""
import numpy as np
import pytest
import torch

from helpers import SMALL_RANGE, SMALL_SHAPE, SMALL_VOXEL, small_points, to_dev
from sparsefusion_amd import spconv

pytestmark = pytest.mark.gpu


def active_set(oracle, seed, n, batch=1, max_voxels=20000):
    coors = []
    for b in range(batch):
        pts = small_points(seed + 17 * b, n + 311 * b)
        _, c, _ = oracle.hard_voxelize(pts, SMALL_VOXEL, SMALL_RANGE, 5, max_voxels)
        coors.append(c)
    return oracle.batch_coors(coors)


def gpu_pairs(indices, batch, shape, ks, st, pd, subm, dev):
    outids, pairs, num = spconv.get_indice_pairs(to_dev(indices, dev), batch, shape, ks, st, pd, 1,
                                                 0, subm)
    torch.cuda.synchronize()
    return outids.cpu().numpy(), pairs.cpu().numpy(), num.cpu().numpy()


CASES = [
    (3, 1, 1, True),
    (3, 2, 1, False),
    (3, 2, (0, 1, 1), False),
    ((3, 1, 1), (2, 1, 1), 0, False),
    (3, 1, 0, False),
    ((1, 3, 3), 1, 0, True),
]


@pytest.mark.parametrize('ks,st,pd,subm', CASES)
@pytest.mark.parametrize('batch', [1, 3])
def test_rulebook_matches_cpu_reference_order(dev, oracle, ks, st, pd, subm, batch):
    idx = active_set(oracle, 3, 2500, batch)
    e_out, e_pairs, e_num, e_shape = oracle.get_indice_pairs(idx, batch, SMALL_SHAPE, ks, st, pd, 1,
                                                             subm)
    outids, pairs, num = gpu_pairs(idx, batch, SMALL_SHAPE, ks, st, pd, subm, dev)
    assert np.array_equal(num, e_num)
    assert np.array_equal(outids, e_out)
    assert np.array_equal(pairs, e_pairs)
    # and the order-free canonical form
    a = oracle.canonical_rulebook(idx, SMALL_SHAPE, e_out, e_shape, e_pairs, e_num)
    b = oracle.canonical_rulebook(idx, SMALL_SHAPE, outids, e_shape, pairs, num)
    assert np.array_equal(a['out_sorted'], b['out_sorted'])
    for x, y in zip(a['pairs'], b['pairs']):
        assert np.array_equal(x, y)


def test_chain_through_all_encoder_rulebooks(dev, oracle):
    """subm1 -> spconv1 -> subm2 -> spconv2 -> subm3 -> spconv3 -> subm4 -> conv_out geometry."""
    idx = active_set(oracle, 11, 6000, 2)
    shape = list(SMALL_SHAPE)
    steps = [(3, 1, 1, True), (3, 2, 1, False), (3, 1, 1, True), (3, 2, 1, False), (3, 1, 1, True),
             (3, 2, (0, 1, 1), False), (3, 1, 1, True), ((3, 1, 1), (2, 1, 1), 0, False)]
    for ks, st, pd, subm in steps:
        e_out, e_pairs, e_num, e_shape = oracle.get_indice_pairs(idx, 2, shape, ks, st, pd, 1, subm)
        outids, pairs, num = gpu_pairs(idx, 2, shape, ks, st, pd, subm, dev)
        assert np.array_equal(outids, e_out) and np.array_equal(num, e_num)
        assert np.array_equal(pairs, e_pairs)
        idx, shape = e_out, e_shape
    assert idx.shape[0] > 0


def test_empty_and_single(dev, oracle):
    empty = np.zeros((0, 4), np.int32)
    outids, pairs, num = gpu_pairs(empty, 1, SMALL_SHAPE, 3, 2, 1, False, dev)
    assert outids.shape == (0, 4) and num.sum() == 0
    one = np.asarray([[0, 0, 0, 0]], np.int32)
    for ks, st, pd, subm in CASES:
        e_out, e_pairs, e_num, _ = oracle.get_indice_pairs(one, 1, SMALL_SHAPE, ks, st, pd, 1, subm)
        outids, pairs, num = gpu_pairs(one, 1, SMALL_SHAPE, ks, st, pd, subm, dev)
        assert np.array_equal(outids, e_out) and np.array_equal(pairs, e_pairs)
        assert np.array_equal(num, e_num)
    corner = np.asarray([[0, 40, 159, 159], [0, 40, 159, 158], [0, 39, 159, 159]], np.int32)
    e_out, e_pairs, e_num, _ = oracle.get_indice_pairs(corner, 1, SMALL_SHAPE, 3, 2, 1, 1, False)
    outids, pairs, num = gpu_pairs(corner, 1, SMALL_SHAPE, 3, 2, 1, False, dev)
    assert np.array_equal(outids, e_out) and np.array_equal(pairs, e_pairs)


def test_duplicate_coordinates_do_not_crash(dev):
    # the reference's own smoke test feeds heavily duplicated coords
    # (tests/test_models/test_common_modules/test_middle_encoders.py:23); behaviour is undefined
    # there; here the first occurrence wins and nothing faults.
    g = torch.Generator().manual_seed(0)
    idx = torch.randint(0, 4, (5000, 4), generator=g, dtype=torch.int32)
    outids, pairs, num = gpu_pairs(idx.numpy(), 4, SMALL_SHAPE, 3, 2, 1, False, dev)
    assert outids.shape[0] <= 4 * 3 * 3 * 3
    outids, pairs, num = gpu_pairs(idx.numpy(), 4, SMALL_SHAPE, 3, 1, 1, True, dev)
    assert outids.shape[0] == 5000

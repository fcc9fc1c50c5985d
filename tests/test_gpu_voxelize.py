"""Hard / dynamic voxelization on the GPU vs the CPU oracle: bit-exact (integer and copy work).

Mirrors tests/test_models/test_voxel_encoder/test_voxelize.py:15-83 of the reference (hard
voxelization equal to the sequential statement; dynamic voxelization membership) plus its
known-answer test (test_voxel_generator.py:6-22), on synthetic inputs."""
import numpy as np
import pytest
import torch

from helpers import SMALL_RANGE, SMALL_VOXEL, small_points, to_dev
from sparsefusion_amd import configs, synth
from sparsefusion_amd.voxelize import Voxelization

pytestmark = pytest.mark.gpu


def run_gpu(points, vs, rng, mp, mv, dev):
    layer = Voxelization(vs, rng, mp, mv).eval()
    v, c, n = layer(to_dev(points, dev))
    torch.cuda.synchronize()
    return v.cpu().numpy(), c.cpu().numpy(), n.cpu().numpy()


def check_equal(points, vs, rng, mp, mv, dev, oracle):
    ev, ec, en = oracle.hard_voxelize(points, vs, rng, mp, mv)
    v, c, n = run_gpu(points, vs, rng, mp, mv, dev)
    assert c.shape == ec.shape, (c.shape, ec.shape)
    assert np.array_equal(c, ec)
    assert np.array_equal(n, en)
    assert np.array_equal(v.view(np.uint32), ev.view(np.uint32))  # bit-exact incl. zero padding
    return ec.shape[0]


def test_known_answer(dev, oracle):
    # tests/test_models/test_voxel_encoder/test_voxel_generator.py:6-22
    np.random.seed(0)
    pts = np.random.rand(1000, 4).astype(np.float32)
    v, c, n = run_gpu(pts, [0.5, 0.5, 0.5], [0, -40, -3, 70.4, 40, 1], 1000, 20000, dev)
    assert c.tolist() == [[7, 81, 1], [6, 81, 0], [7, 80, 1], [6, 81, 1], [7, 81, 0], [6, 80, 1],
                          [7, 80, 0], [6, 80, 0]]
    assert n.tolist() == [120, 121, 127, 134, 115, 127, 125, 131]
    assert v.shape == (8, 1000, 4)
    check_equal(pts, [0.5, 0.5, 0.5], [0, -40, -3, 70.4, 40, 1], 1000, 20000, dev, oracle)


@pytest.mark.parametrize('n,mp,mv', [(3000, 10, 20000), (4096, 3, 512), (5000, 1, 100),
                                     (2000, 5, 1), (777, 35, 20000), (64, 10, 0)])
def test_small_frames_caps(dev, oracle, n, mp, mv):
    pts = small_points(n, n)
    check_equal(pts, SMALL_VOXEL, SMALL_RANGE, mp, mv, dev, oracle)


@pytest.mark.parametrize('mp,mv', [(10, 20000), (5, 20000), (3, 20000), (20, 20000), (10, 40), (1, 20000)])
def test_crowded_voxels(dev, oracle, mp, mv):
    """A few voxels swallow most of the cloud (a return right at the sensor): long runs of
    consecutive points in one voxel, two voxels alternating point by point, runs broken by strays,
    and the same voxels again later in the scan.  The run / bound shortcuts of vox_insert and
    vox_collect must keep exactly the first max_points points of each voxel, in scan order."""
    rs = np.random.RandomState(77)
    base = small_points(9, 6000)
    cells = np.array([[0.3, 0.3, 0.1], [-2.1, 1.7, 0.5], [3.3, -0.9, -0.7]], np.float32)

    def near(ci, n):      # n points inside the voxel of cells[ci]
        p = np.zeros((n, base.shape[1]), np.float32)
        p[:, :3] = cells[ci] + rs.uniform(-0.01, 0.01, (n, 3)).astype(np.float32)
        p[:, 3:] = rs.rand(n, base.shape[1] - 3).astype(np.float32)
        return p
    alt = np.empty((600, base.shape[1]), np.float32)
    alt[0::2], alt[1::2] = near(1, 300), near(2, 300)
    broken = near(0, 900)
    broken[::7] = base[:129]                       # a stray point every 7th slot of a run
    pts = np.concatenate([base[:500], near(0, 3000), base[500:900], alt, base[900:2500], broken,
                          near(1, 70), base[2500:], near(2, 5000)])
    check_equal(pts, SMALL_VOXEL, SMALL_RANGE, mp, mv, dev, oracle)
    check_equal(pts[::-1].copy(), SMALL_VOXEL, SMALL_RANGE, mp, mv, dev, oracle)


def test_edge_values(dev, oracle):
    pts = small_points(5, 600)
    pts[10, 0] = np.nan
    pts[11, 1] = np.inf
    pts[12, 2] = -np.inf
    pts[13, 0] = 8.0                      # exactly on the upper bound: excluded
    pts[14, 0] = -8.0                     # exactly on the lower bound: included
    pts[15, :3] = [7.9999995, 7.9999995, 1.9999999]
    pts[16, 0] = 3.0e38
    pts[17, 0] = np.nextafter(np.float32(-8.0), np.float32(-9.0))
    pts[18:40, :3] = pts[18, :3]          # many points in one voxel
    check_equal(pts, SMALL_VOXEL, SMALL_RANGE, 4, 20000, dev, oracle)
    # dynamic path agrees with the oracle coordinate by coordinate (all -1 on rejection)
    layer = Voxelization(SMALL_VOXEL, SMALL_RANGE, -1, 20000)
    c = layer(to_dev(pts, dev)).cpu().numpy()
    assert np.array_equal(c, oracle.dynamic_voxelize(pts, SMALL_VOXEL, SMALL_RANGE))


def test_empty_and_all_rejected(dev, oracle):
    empty = np.zeros((0, 5), np.float32)
    v, c, n = run_gpu(empty, SMALL_VOXEL, SMALL_RANGE, 10, 100, dev)
    assert v.shape == (0, 10, 5) and c.shape == (0, 3) and n.shape == (0,)
    far = np.full((50, 5), 1000.0, np.float32)
    v, c, n = run_gpu(far, SMALL_VOXEL, SMALL_RANGE, 10, 100, dev)
    assert v.shape[0] == 0


def test_cut_drops_points_of_existing_voxels(dev, oracle):
    # voxel A, voxel B, (cap=2) voxel C -> break; the later point of voxel A must be dropped
    pts = np.zeros((5, 5), np.float32)
    pts[:, :3] = [[0.05, 0.05, 0.05], [1.05, 0.05, 0.05], [0.06, 0.06, 0.06],
                  [2.05, 0.05, 0.05], [0.07, 0.07, 0.07]]
    ev, ec, en = oracle.hard_voxelize(pts, SMALL_VOXEL, SMALL_RANGE, 10, 2)
    assert en.tolist() == [2, 1]
    check_equal(pts, SMALL_VOXEL, SMALL_RANGE, 10, 2, dev, oracle)


def test_nuscenes_frame_full_size(dev, oracle):
    cfg = configs.NUSC['pts_voxel_layer']
    pts = synth.nuscenes_frame(0)
    m = check_equal(pts, cfg['voxel_size'], cfg['point_cloud_range'], cfg['max_num_points'],
                    cfg['max_voxels'][1], dev, oracle)
    assert m > 50000
    # train-pipeline variant: filtered + shuffled, training cap
    tp = synth.train_variant(pts, cfg['point_cloud_range'], 7)
    check_equal(tp, cfg['voxel_size'], cfg['point_cloud_range'], cfg['max_num_points'], 60000,
                dev, oracle)


def test_waymo_stress_cap_triggers(dev, oracle):
    cfg = configs.WAYMO['pts_voxel_layer']
    pts = synth.waymo_frame(0)
    m = check_equal(pts, cfg['voxel_size'], cfg['point_cloud_range'], cfg['max_num_points'],
                    cfg['max_voxels'], dev, oracle)
    assert m == cfg['max_voxels']

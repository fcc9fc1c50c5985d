"""extract_voxel_heights (detectors/sparsefusion.py:73-112).  The reference holds no fixture for it,
so the oracle is pinned by a line-by-line numpy transcription of those lines on small inputs
(parity unpinned by upstream data); the GPU kernel is then compared with the oracle, exactly
(min / max only select values)."""
import numpy as np
import pytest
import torch


def transcription(voxels, coors, grid_size, osf):
    """sparsefusion.py:73-112 with numpy in place of torch, statement by statement."""
    batch_size = int(coors[-1, 0]) + 1
    height_num = grid_size[2]
    x_num, y_num = grid_size[0] // osf, grid_size[1] // osf
    v = voxels[:, :, 2].copy()
    v[v == 0] = 100
    min_voxel = v.min(-1)
    v[v == 100] = -200
    max_voxel = v.max(-1)
    mn = np.zeros((batch_size, y_num, x_num, osf * osf), np.float32) + 100
    mx = np.zeros((batch_size, y_num, x_num, osf * osf), np.float32) - 200
    b, h = coors[:, 0], coors[:, 1]
    yi, xi, yo, xo = coors[:, 2] // osf, coors[:, 3] // osf, coors[:, 2] % osf, coors[:, 3] % osf
    for hid in range(height_num):
        k = h == hid
        sel = (b[k], yi[k], xi[k], yo[k] * osf + xo[k])
        mn[sel] = np.minimum(mn[sel], min_voxel[k])
        mx[sel] = np.maximum(mx[sel], max_voxel[k])
    return mn.min(-1), mx.max(-1)


def make_case(seed, m, batch, grid=(32, 24, 6), mp=5):
    rng = np.random.default_rng(seed)
    cells = rng.choice(batch * grid[2] * grid[1] * grid[0], size=m, replace=False)
    cells.sort()                                      # batch-sorted like the detector's voxelize()
    b, r = np.divmod(cells, grid[2] * grid[1] * grid[0])
    z, r = np.divmod(r, grid[1] * grid[0])
    y, x = np.divmod(r, grid[0])
    coors = np.stack([b, z, y, x], 1).astype(np.int32)
    voxels = rng.normal(0, 2, (m, mp, 5)).astype(np.float32)
    npts = rng.integers(1, mp + 1, m)
    for i in range(m):
        voxels[i, npts[i]:] = 0                       # zero padding slots
    voxels[rng.random(m) < 0.05, 0, 2] = 0.0          # a real point with z == 0 counts as padding
    return voxels, coors, list(grid)


@pytest.mark.parametrize('seed,m,batch,osf', [(0, 900, 1, 4), (1, 1500, 3, 8), (2, 40, 2, 2)])
def test_oracle_matches_transcription(oracle, seed, m, batch, osf):
    voxels, coors, grid = make_case(seed, m, batch)
    emn, emx = transcription(voxels, coors, grid, osf)
    b = int(coors[-1, 0]) + 1
    mn, mx = oracle.voxel_heights(voxels, coors, b, grid, osf)
    assert np.array_equal(mn, emn) and np.array_equal(mx, emx)
    if m < 100:
        assert (mn == 100).any() and (mx == -200).any()   # empty cells keep the sentinels


@pytest.mark.gpu
@pytest.mark.parametrize('seed,m,batch,osf', [(0, 900, 1, 4), (1, 1500, 3, 8), (2, 40, 2, 2)])
def test_gpu_matches_oracle_small(dev, oracle, seed, m, batch, osf):
    from sparsefusion_amd.voxel_heights import extract_voxel_heights
    voxels, coors, grid = make_case(seed, m, batch)
    b = int(coors[-1, 0]) + 1
    emn, emx = oracle.voxel_heights(voxels, coors, b, grid, osf)
    mn, mx = extract_voxel_heights(torch.from_numpy(voxels).to(dev), torch.from_numpy(coors).to(dev),
                                   grid, osf)
    assert np.array_equal(mn.cpu().numpy(), emn) and np.array_equal(mx.cpu().numpy(), emx)


@pytest.mark.gpu
def test_gpu_full_frame(dev, oracle):
    from sparsefusion_amd import configs, synth
    from sparsefusion_amd.voxel_heights import extract_voxel_heights
    from sparsefusion_amd.voxelize import Voxelization
    vl = configs.NUSC['pts_voxel_layer']
    layer = Voxelization(vl['voxel_size'], vl['point_cloud_range'], vl['max_num_points'],
                         vl['max_voxels']).eval()
    v, c, n = layer(torch.from_numpy(synth.nuscenes_frame(2)).to(dev))
    c4 = torch.nn.functional.pad(c, (1, 0), value=0)
    grid = [1440, 1440, 40]
    mn, mx = extract_voxel_heights(v, c4.contiguous(), grid, 8)
    emn, emx = oracle.voxel_heights(v.cpu().numpy(), c4.cpu().numpy(), 1, grid, 8)
    assert mn.shape == (1, 180, 180)
    assert np.array_equal(mn.cpu().numpy(), emn) and np.array_equal(mx.cpu().numpy(), emx)

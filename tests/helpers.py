"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np
import torch

from sparsefusion_amd import configs, synth

SMALL_RANGE = [-8.0, -8.0, -2.0, 8.0, 8.0, 2.0]
SMALL_VOXEL = [0.1, 0.1, 0.1]            # grid 160 x 160 x 40 -> sparse_shape [41,160,160]
SMALL_SHAPE = [41, 160, 160]


def small_points(seed, n, nfeat=5):
    return synth.small_frame(seed, n, SMALL_RANGE, spread=3.0, nfeat=nfeat)


def to_dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def encoder_cfg(name='nusc'):
    cfg = dict((configs.NUSC if name == 'nusc' else configs.WAYMO)['pts_middle_encoder'])
    cfg.pop('type')
    return cfg


def weight_cfg(cfg):
    return {k: cfg[k] for k in ('base_channels', 'output_channels', 'encoder_channels',
                                'block_type') if k in cfg}


def torch_state_dict(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def assert_close(got, expect, rtol=1e-4, atol_scale=1e-5):
    """|got - expect| <= rtol*|expect| + atol_scale*max|expect| elementwise (fp32 sums in a
    different order: the absolute term covers cancellation near zero)."""
    got = np.asarray(got, np.float64)
    expect = np.asarray(expect, np.float64)
    assert got.shape == expect.shape, (got.shape, expect.shape)
    tol = rtol * np.abs(expect) + atol_scale * max(np.abs(expect).max(), 1e-30)
    bad = np.abs(got - expect) > tol
    assert not bad.any(), (int(bad.sum()), float(np.abs(got - expect).max()),
                           float(np.abs(expect).max()))


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))

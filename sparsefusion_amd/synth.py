"""Seeded synthetic LiDAR frames and SparseEncoder weights (there is no dataset or checkpoint
on the build or GPU boxes).  Used by bench.py, the parity tests and the fixture generator.

Frame layout follows the reference's loader contract, float32 `[x, y, z, intensity, dt]`,
key-frame points first and then the older sweeps in order
(mmdet3d/datasets/pipelines/loading.py:607-655).  The generator is a small ray caster:
`beams x azimuths` rays per sweep against a ground plane plus random boxes (vehicles) and
walls, the ego moving 0.5 m per sweep, Gaussian range noise.  Frame i uses seed 1000 + i.

Weights: seeded He-uniform (see _he_uniform for why not the reference init), stored in the
spconv-2 layout `(Cout, kD, kH, kW, Cin)` the reference's checkpoints use
(mmdet3d/ops/spconv/overwrite_spconv/write_spconv2.py:46-47); BN affine / running stats
random so the folded epilogue is exercised.
"""
import math

import numpy as np


def _scene(rng, extent, n_cars, n_walls, ground_z, wall_min_dist=12.0):
    lo, hi = [], []
    for _ in range(n_cars):
        while True:
            c = rng.uniform(-extent, extent, 2)
            if np.hypot(*c) > 4.0:
                break
        yaw_swap = rng.random() < 0.5
        sx, sy = (4.6, 1.9) if not yaw_swap else (1.9, 4.6)
        sx *= rng.uniform(0.8, 2.0)
        sy *= rng.uniform(0.8, 1.3)
        sz = rng.uniform(1.4, 3.2)
        lo.append([c[0] - sx / 2, c[1] - sy / 2, ground_z])
        hi.append([c[0] + sx / 2, c[1] + sy / 2, ground_z + sz])
    for _ in range(n_walls):
        while True:
            c = rng.uniform(-extent, extent, 2)
            if np.hypot(*c) > wall_min_dist:
                break
        along_x = rng.random() < 0.5
        ln, th = rng.uniform(10, 40), rng.uniform(0.4, 3.0)
        sx, sy = (ln, th) if along_x else (th, ln)
        sz = rng.uniform(3.0, 12.0)
        lo.append([c[0] - sx / 2, c[1] - sy / 2, ground_z])
        hi.append([c[0] + sx / 2, c[1] + sy / 2, ground_z + sz])
    return np.asarray(lo, np.float32), np.asarray(hi, np.float32)


def _cast(origin, dirs, lo, hi, ground_z, max_range):
    """Nearest hit distance per ray (inf where nothing within max_range); slab test per box."""
    with np.errstate(divide='ignore', invalid='ignore'):
        dx, dy, dz = (np.ascontiguousarray(dirs[:, i]) for i in range(3))
        t = np.where(dz < 0, (ground_z - origin[2]) / dz, np.inf).astype(np.float32)
        ix, iy, iz = (1.0 / dx).astype(np.float32), (1.0 / dy).astype(np.float32), \
            (1.0 / dz).astype(np.float32)
        for b in range(lo.shape[0]):
            ax, bx = (lo[b, 0] - origin[0]) * ix, (hi[b, 0] - origin[0]) * ix
            ay, by = (lo[b, 1] - origin[1]) * iy, (hi[b, 1] - origin[1]) * iy
            az, bz = (lo[b, 2] - origin[2]) * iz, (hi[b, 2] - origin[2]) * iz
            tmin = np.maximum(np.maximum(np.minimum(ax, bx), np.minimum(ay, by)),
                              np.minimum(az, bz))
            tmax = np.minimum(np.minimum(np.maximum(ax, bx), np.maximum(ay, by)),
                              np.maximum(az, bz))
            tmin = np.maximum(tmin, 0.0)
            t = np.where((tmax >= tmin) & (tmin < t), tmin, t)
    return np.where(t <= max_range, t, np.inf)


def lidar_frame(seed, sweeps, beams, azimuths, elev_deg, ground_z, max_range, extent,
                n_cars, n_walls, ego_step=0.5, sigma=0.02, dt_step=0.05, clutter_p=0.28,
                wall_min_dist=18.0):
    rng = np.random.default_rng(seed)
    lo, hi = _scene(rng, extent, n_cars, n_walls, ground_z, wall_min_dist)
    elev0 = np.deg2rad(np.linspace(elev_deg[0], elev_deg[1], beams)).astype(np.float32)
    out = []
    for s in range(sweeps):
        az = (np.arange(azimuths, dtype=np.float32) + rng.random(dtype=np.float32)) * \
            np.float32(2 * math.pi / azimuths)
        # ego pose jitter per sweep (pitch, height, lateral drift): rings do not coincide
        elev = elev0 + np.float32(np.deg2rad(rng.normal(0, 0.35)))
        origin = np.asarray([-ego_step * s, rng.normal(0, 0.15) * (s > 0),
                             rng.normal(0, 0.03) * (s > 0)], np.float32)
        # firing order: azimuth-major, beams inside (a spinning LiDAR's packet order)
        ce, se = np.cos(elev)[None, :], np.sin(elev)[None, :]
        d = np.stack([ce * np.cos(az)[:, None], ce * np.sin(az)[:, None],
                      np.broadcast_to(se, (azimuths, beams))], -1).reshape(-1, 3).astype(np.float32)
        t = _cast(origin, d, lo, hi, ground_z, max_range)
        # vegetation / clutter: some of the shallow beams return early at a random range
        clutter = (rng.random(t.shape[0]) < clutter_p) & (d[:, 2] > -0.14)
        tc = rng.uniform(4.0, 0.75 * max_range, t.shape[0]).astype(np.float32)
        t = np.where(clutter & (tc < t), tc, t)
        ok = np.isfinite(t)
        t = t[ok] + rng.normal(0.0, sigma, int(ok.sum())).astype(np.float32)
        p = origin[None] + d[ok] * t[:, None]
        inten = rng.uniform(0, 255, p.shape[0]).astype(np.float32)
        dt = np.full(p.shape[0], dt_step * s, np.float32)
        out.append(np.concatenate([p, inten[:, None], dt[:, None]], 1).astype(np.float32))
    return np.ascontiguousarray(np.concatenate(out, 0), np.float32)


def nuscenes_frame(index=0):
    """nuScenes-shaped 10-sweep frame (32 beams x 1084 azimuths), test-pipeline style:
    unshuffled, out-of-range points left in (configs/sparsefusion_nusc_voxel_LC_r50.py:58-93)."""
    return lidar_frame(1000 + index, sweeps=10, beams=32, azimuths=1084,
                       elev_deg=(-30.67, 10.67), ground_z=-1.84, max_range=75.0, extent=60.0,
                       n_cars=90, n_walls=40)


def waymo_frame(index=0):
    """Waymo-shaped stress frame: 64 beams x 2650 azimuths x 6 returns/sweeps (~1 M points)."""
    return lidar_frame(5000 + index, sweeps=6, beams=64, azimuths=2650,
                       elev_deg=(-17.6, 2.4), ground_z=-1.7, max_range=85.0, extent=80.0,
                       n_cars=140, n_walls=110, ego_step=0.3)


def train_variant(points, point_cloud_range, seed):
    """Train-pipeline variant: strict range filter (core/points/base_points.py:159-164) and
    shuffle (datasets/pipelines/transforms_3d.py:748-762)."""
    r = np.asarray(point_cloud_range, np.float32)
    m = ((points[:, 0] > r[0]) & (points[:, 1] > r[1]) & (points[:, 2] > r[2]) &
         (points[:, 0] < r[3]) & (points[:, 1] < r[4]) & (points[:, 2] < r[5]))
    p = points[m]
    rng = np.random.default_rng(seed)
    return np.ascontiguousarray(p[rng.permutation(p.shape[0])])


def small_frame(seed, n, point_cloud_range, spread=6.0, nfeat=5):
    """A few thousand points clustered on a tilted plane + blobs inside the range, dense
    enough that sparse-conv neighbours exist at every stage; for oracle-sized parity tests."""
    rng = np.random.default_rng(seed)
    r = np.asarray(point_cloud_range, np.float32)
    c = (r[:3] + r[3:]) / 2
    xy = rng.normal(0, spread / 2, (n, 2)).astype(np.float32)
    z = (0.05 * xy[:, 0] - 0.03 * xy[:, 1] + rng.normal(0, 0.15, n)).astype(np.float32)
    blob = rng.random(n) < 0.3
    z = np.where(blob, rng.uniform(-1.0, 1.5, n), z).astype(np.float32)
    p = np.concatenate([xy + c[None, :2], (z + c[2])[:, None]], 1)
    extra = [rng.uniform(0, 255, n), rng.integers(0, 10, n) * 0.05][:max(nfeat - 3, 0)]
    cols = [p] + [e[:, None] for e in extra]
    return np.ascontiguousarray(np.concatenate(cols, 1), np.float32)


# ------------------------------------------------------------------ weights

def _he_uniform(rng, shape_k_ci_co):
    """He-uniform with an effective fan-in of 9*Cin (about the mean number of active
    neighbours x Cin on LiDAR surfaces).  The reference's own init -- kaiming_uniform_(a=sqrt5)
    on the (kD,kH,kW,Cin,Cout) tensor, conv.py:106-107 -- yields fan_in = kH*kW*Cin*Cout and,
    with random eval-mode BN statistics, activations that vanish within a few layers; this
    keeps every layer's output O(1) so the parity tests compare real magnitudes."""
    kd, kh, kw, ci, co = shape_k_ci_co
    bound = math.sqrt(3.0 / (min(kd * kh * kw, 9) * ci))
    return rng.uniform(-bound, bound, shape_k_ci_co).astype(np.float32)


def encoder_conv_list(in_channels, base_channels=16, output_channels=128,
                      encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                      block_type='conv_module', **_):
    """[(weight key, bn prefix, ksize, cin, cout)] for every conv of a SparseEncoder, in
    forward order, under the reference's state-dict names."""
    out = [('conv_input.0.weight', 'conv_input.1', (3, 3, 3), in_channels, base_channels)]
    cin = base_channels
    for i, blocks in enumerate(encoder_channels):
        for j, cout in enumerate(tuple(blocks)):
            p = f'encoder_layers.encoder_layer{i + 1}.{j}'
            is_down = (block_type == 'basicblock' and j == len(blocks) - 1 and
                       i != len(encoder_channels) - 1)
            if block_type == 'basicblock' and not is_down:
                out.append((f'{p}.conv1.weight', f'{p}.bn1', (3, 3, 3), cout, cout))
                out.append((f'{p}.conv2.weight', f'{p}.bn2', (3, 3, 3), cout, cout))
            else:
                out.append((f'{p}.0.weight', f'{p}.1', (3, 3, 3), cin, cout))
            cin = cout
    out.append(('conv_out.0.weight', 'conv_out.1', (3, 1, 1), cin, output_channels))
    return out


def encoder_state_dict(seed, in_channels, layout='spconv2', **cfg):
    """Random-init state dict (numpy) for SparseEncoder(in_channels, **cfg)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for wkey, bn, ks, ci, co in encoder_conv_list(in_channels, **cfg):
        w = _he_uniform(rng, ks + (ci, co))
        if layout == 'spconv2':
            w = np.ascontiguousarray(np.transpose(w, (4, 0, 1, 2, 3)))
        sd[wkey] = w
        sd[f'{bn}.weight'] = rng.uniform(0.5, 1.5, co).astype(np.float32)
        sd[f'{bn}.bias'] = rng.normal(0, 0.1, co).astype(np.float32)
        sd[f'{bn}.running_mean'] = rng.normal(0, 0.1, co).astype(np.float32)
        sd[f'{bn}.running_var'] = rng.uniform(0.5, 1.5, co).astype(np.float32)
        sd[f'{bn}.num_batches_tracked'] = np.asarray(0, np.int64)
    return sd


def hard_vfe_state_dict(seed, in_channels, feat_channels):
    """Random-init state dict (numpy) for HardVFE(in_channels (decorated), feat_channels) under
    the reference's key names (vfe_layers.N.linear.weight, vfe_layers.N.norm.*)."""
    rng = np.random.default_rng(seed)
    sd = {}
    chans = [in_channels] + list(feat_channels)
    for i in range(len(chans) - 1):
        cin = chans[i] * (2 if i > 0 else 1)
        co = chans[i + 1]
        sd[f'vfe_layers.{i}.linear.weight'] = (rng.uniform(-1, 1, (co, cin)) *
                                               math.sqrt(3.0 / cin)).astype(np.float32)
        sd[f'vfe_layers.{i}.norm.weight'] = rng.uniform(0.5, 1.5, co).astype(np.float32)
        sd[f'vfe_layers.{i}.norm.bias'] = rng.normal(0, 0.1, co).astype(np.float32)
        sd[f'vfe_layers.{i}.norm.running_mean'] = rng.normal(0, 0.1, co).astype(np.float32)
        sd[f'vfe_layers.{i}.norm.running_var'] = rng.uniform(0.5, 1.5, co).astype(np.float32)
        sd[f'vfe_layers.{i}.norm.num_batches_tracked'] = np.asarray(0, np.int64)
    return sd


def _bn2d_state(rng, sd, prefix, co):
    sd[f'{prefix}.weight'] = rng.uniform(0.5, 1.5, co).astype(np.float32)
    sd[f'{prefix}.bias'] = rng.normal(0, 0.1, co).astype(np.float32)
    sd[f'{prefix}.running_mean'] = rng.normal(0, 0.1, co).astype(np.float32)
    sd[f'{prefix}.running_var'] = rng.uniform(0.5, 1.5, co).astype(np.float32)
    sd[f'{prefix}.num_batches_tracked'] = np.asarray(0, np.int64)


def second_state_dict(seed, in_channels, out_channels, layer_nums, **_):
    """Random-init state dict (numpy) for SECOND under the reference's key names
    (blocks.I.{0,3,..}.weight, blocks.I.{1,4,..}.*): He-uniform filters, BN statistics as in
    encoder_state_dict."""
    rng = np.random.default_rng(seed)
    sd = {}
    cins = [in_channels, *out_channels[:-1]]
    for i, n in enumerate(layer_nums):
        for j in range(n + 1):
            ci, co = (cins[i] if j == 0 else out_channels[i]), out_channels[i]
            bound = math.sqrt(6.0 / (9 * ci))
            sd[f'blocks.{i}.{3 * j}.weight'] = rng.uniform(-bound, bound, (co, ci, 3, 3)).astype(np.float32)
            _bn2d_state(rng, sd, f'blocks.{i}.{3 * j + 1}', co)
    return sd


def second_fpn_state_dict(seed, in_channels, out_channels, upsample_strides,
                          use_conv_for_no_stride=False, **_):
    """Random-init state dict (numpy) for SECONDFPN (deblocks.I.0.weight, deblocks.I.1.*)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for i, (ci, co, s) in enumerate(zip(in_channels, out_channels, upsample_strides)):
        if s > 1 or (s == 1 and not use_conv_for_no_stride):
            shape = (ci, co, s, s)                      # nn.ConvTranspose2d
        else:
            k = int(np.round(1 / s))
            shape = (co, ci, k, k)                      # nn.Conv2d
        bound = math.sqrt(6.0 / ci)
        sd[f'deblocks.{i}.0.weight'] = rng.uniform(-bound, bound, shape).astype(np.float32)
        _bn2d_state(rng, sd, f'deblocks.{i}.1', co)
    return sd


def pillar_feature_net_state_dict(seed, in_channels, feat_channels):
    """Random-init state dict (numpy) for PillarFeatureNet(in_channels (decorated), feat_channels)
    under the reference's key names (pfn_layers.N.linear.weight, pfn_layers.N.norm.*)."""
    rng = np.random.default_rng(seed)
    sd = {}
    chans = [in_channels] + list(feat_channels)
    for i in range(len(chans) - 1):
        last = i == len(chans) - 2
        cin, co = chans[i], (chans[i + 1] if last else chans[i + 1] // 2)
        sd[f'pfn_layers.{i}.linear.weight'] = (rng.uniform(-1, 1, (co, cin)) *
                                               math.sqrt(3.0 / cin)).astype(np.float32)
        _bn2d_state(rng, sd, f'pfn_layers.{i}.norm', co)
    return sd


def sparse_unet_state_dict(seed, in_channels, base_channels=16, output_channels=128,
                           encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                           decoder_channels=((64, 64, 64), (64, 64, 32), (32, 32, 16), (16, 16, 16)),
                           **_):
    """Random-init state dict (numpy, spconv-2 filter layout) for SparseUNet under the reference's
    key names: the conv_module encoder plus lateral_layerN.{conv1,bn1,conv2,bn2},
    merge_layerN.{0,1}, upsample_layerN.{0,1} (sparse_unet.py:237-292)."""
    sd = encoder_state_dict(seed, in_channels, base_channels=base_channels,
                            output_channels=output_channels, encoder_channels=encoder_channels,
                            block_type='conv_module')
    rng = np.random.default_rng(seed + 1)

    def conv(wkey, bn, ci, co):
        w = _he_uniform(rng, (3, 3, 3, ci, co))
        sd[wkey] = np.ascontiguousarray(np.transpose(w, (4, 0, 1, 2, 3)))
        _bn2d_state(rng, sd, bn, co)

    cin = tuple(encoder_channels[-1])[-1]
    n = len(decoder_channels)
    for i, (c_lat, c_merge, c_up) in enumerate(decoder_channels):
        lv = n - i
        conv(f'lateral_layer{lv}.conv1.weight', f'lateral_layer{lv}.bn1', cin, c_lat)
        conv(f'lateral_layer{lv}.conv2.weight', f'lateral_layer{lv}.bn2', c_lat, c_lat)
        conv(f'merge_layer{lv}.0.weight', f'merge_layer{lv}.1', cin * 2, c_merge)
        conv(f'upsample_layer{lv}.0.weight', f'upsample_layer{lv}.1', cin, c_up)
        cin = c_up
    return sd

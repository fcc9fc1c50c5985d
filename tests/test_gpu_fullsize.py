"""Full-size parity for the BASELINE.json configs (the other GPU tests use 41x160x160):

* configs[1]  nuScenes-shaped frame 0, fp32, 41x1440x1440 (32-bit keys over 85 M cells, 2^19..2^21
              slot hash tables, padded neighbour-table strides, the 520-tile balanced conv path);
* configs[2]  B = 8 distinct nuScenes-shaped frames, bf16 path;
* configs[4]  Waymo-shaped frame 0, 41x1504x1504, HardVFE 5->64 + 64-channel encoder input, the
              150 000-voxel cap hit.

Each through LidarBackbone (the path bench.py times).  Checked:
  coors / num_points          == oracle voxelizer, bit for bit
  all 8 rulebooks             exported to the reference's (outids, indicePairs, indiceNum) in the CPU
                              reference's order; sha256 == tests/golden/fullsize_rulebooks.json, which
                              tests/golden/make_golden.py wrote FROM THE COMPILED REFERENCE
                              (oracle/_ref/spconv: include/spconv/geometry.h), and == the oracle
  BEV                         fp32: |a-b| <= 1e-4|b| + 1e-5 max|b| vs the oracle
                              (sparse_encoder.py:95-127 contract); bf16: max <= 5e-2 max|b|,
                              RMS <= 2e-2 RMS(b) vs the FP32 oracle (tolerances of test_gpu_bf16.py)
"""
import numpy as np
import pytest
import torch

from helpers import (ENCODER_CHAIN, assert_close, fullsize_golden, rulebook_digest, to_dev,
                     torch_state_dict)
from sparsefusion_amd import configs, spconv, synth
from sparsefusion_amd.pipeline import LidarBackbone
from sparsefusion_amd.sparse_encoder import SparseEncoder

pytestmark = pytest.mark.gpu


def build(cfg, dev, batch, max_points, dtype='f32', seed=1):
    ecfg = dict(cfg['pts_middle_encoder'])
    ecfg.pop('type')
    wcfg = {k: ecfg[k] for k in ('output_channels', 'encoder_channels', 'block_type')}
    sd = synth.encoder_state_dict(seed, ecfg['in_channels'], **wcfg)
    enc = SparseEncoder(**ecfg)
    enc.load_state_dict(torch_state_dict(sd))
    enc = enc.to(dev).eval()
    vfe, vsd = None, None
    vcfg = dict(cfg['pts_voxel_encoder'])
    if vcfg.pop('type') == 'HardVFE':
        from sparsefusion_amd.voxel_encoder import HardVFE
        vfe = HardVFE(**vcfg)
        vsd = synth.hard_vfe_state_dict(seed + 1, vfe.in_channels, vcfg['feat_channels'])
        vfe.load_state_dict(torch_state_dict(vsd))
        vfe = vfe.to(dev).eval()
    bb = LidarBackbone(cfg['pts_voxel_layer'], enc, dev, max_points, batch_size=batch, dtype=dtype,
                       voxel_encoder=vfe)
    okw = dict(encoder_paddings=ecfg['encoder_paddings'], **wcfg)
    return bb, sd, vsd, ecfg, okw


def gpu_rulebook_digests(bb):
    """Every rulebook of the last enqueued batch in the reference's pair format + its digest."""
    counts = bb.counts.tolist()
    out = []
    for rb in bb.rulebooks:
        src, dst = bb.sets[rb['src']], bb.sets[rb['dst']]
        n_in, n_out = counts[rb['src']], counts[rb['dst']]
        cin = spconv.CoordSet(src['indices'][:n_in], src['shape'], bb.B)
        cout = cin if rb['subm'] else spconv.CoordSet(dst['indices'][:n_out], dst['shape'], bb.B)
        r = spconv.Rulebook(rb['nbr'], rb['K'], cin, cout, rb['ks'], rb['st'], rb['pd'], rb['subm'])
        pairs, num = r.indice_pairs()
        torch.cuda.synchronize()
        pairs, num = pairs.cpu().numpy(), num.cpu().numpy()
        outids = cout.indices.cpu().numpy()
        out.append(dict(n_in=n_in, n_out=n_out, pairs=int(num.sum()),
                        sha256=rulebook_digest(outids, pairs, num)))
    return out


def oracle_digests(oracle, coors4, batch, shape):
    out, idx, shape = [], coors4, list(shape)
    for ks, st, pd, subm in ENCODER_CHAIN:
        o, p, n, oshape = oracle.get_indice_pairs(idx, batch, shape, ks, st, pd, 1, subm)
        out.append(dict(n_in=int(idx.shape[0]), n_out=int(o.shape[0]), pairs=int(n.sum()),
                        sha256=rulebook_digest(o, p, n)))
        idx, shape = o, oshape
    return out


def voxelize_oracle(oracle, cfg, frames):
    vl = cfg['pts_voxel_layer']
    mv = vl['max_voxels']
    mv = mv[1] if isinstance(mv, (tuple, list)) else mv
    vs, cs, ns = [], [], []
    for p in frames:
        v, c, n = oracle.hard_voxelize(p, vl['voxel_size'], vl['point_cloud_range'],
                                       vl['max_num_points'], mv)
        vs.append(v), cs.append(c), ns.append(n)
    return vs, cs, ns


def rms(x):
    return float(np.sqrt(np.mean(np.square(np.asarray(x, np.float64)))))


def test_nuscenes_frame_fp32(dev, oracle):
    """BASELINE configs[1]."""
    cfg = configs.NUSC
    frames = [synth.nuscenes_frame(0)]
    bb, sd, _, ecfg, okw = build(cfg, dev, 1, frames[0].shape[0])
    with torch.no_grad():
        bev = bb([to_dev(p, dev) for p in frames]).clone()
    res = bb.results()
    assert res['overflow'] == 0
    vs, cs, ns = voxelize_oracle(oracle, cfg, frames)
    coors4 = oracle.batch_coors(cs)
    assert np.array_equal(res['coors'].cpu().numpy(), coors4)
    assert np.array_equal(res['num_points'].cpu().numpy(), ns[0])
    assert np.array_equal(res['voxels'].cpu().numpy(), vs[0])
    gold = fullsize_golden()
    got = gpu_rulebook_digests(bb)
    assert got == gold['nusc_frame0']                      # == the compiled reference
    assert got == oracle_digests(oracle, coors4, 1, ecfg['sparse_shape'])
    feats = oracle.vfe_mean(vs[0], ns[0], 5)
    expect = oracle.sparse_encoder_forward(sd, feats, coors4, 1, ecfg['sparse_shape'], 5, **okw)
    assert expect.shape == (1, 256, 180, 180)
    assert_close(bev.cpu().numpy(), expect, 1e-4)
    g = gold['nusc_frame0_features']                       # same weights (seed 1): regression pin
    assert abs(np.abs(expect).mean() - g['bev_abs_mean']) <= 1e-5 * g['bev_abs_mean']
    assert int(np.count_nonzero(expect)) == g['bev_nonzero']


@pytest.fixture(scope='module')
def batch8(oracle):
    """The 8-frame batch of configs[2] through the fp32 oracle, once for both B = 8 tests (same
    frames, same seeded weights: build(seed=1))."""
    cfg = configs.NUSC
    frames = [synth.nuscenes_frame(i) for i in range(8)]
    ecfg = dict(cfg['pts_middle_encoder'])
    ecfg.pop('type')
    wcfg = {k: ecfg[k] for k in ('output_channels', 'encoder_channels', 'block_type')}
    sd = synth.encoder_state_dict(1, ecfg['in_channels'], **wcfg)
    vs, cs, ns = voxelize_oracle(oracle, cfg, frames)
    coors4 = oracle.batch_coors(cs)
    feats = np.concatenate([oracle.vfe_mean(v, n, 5) for v, n in zip(vs, ns)], 0)
    expect = oracle.sparse_encoder_forward(sd, feats, coors4, 8, ecfg['sparse_shape'], 5,
                                           encoder_paddings=ecfg['encoder_paddings'], **wcfg)
    return dict(frames=frames, ns=ns, coors4=coors4, expect=expect)


def test_nuscenes_batch8_fp32(dev, batch8):
    """Eight full-size frames per batch on the fp32 path (`bench.py --batch 8 --dtype f32`): about
    16 tiles per persistent workgroup in the 64- and 128-channel layers, 1.1 M rows at stage 2."""
    cfg = configs.NUSC
    frames = batch8['frames']
    bb, sd, _, ecfg, okw = build(cfg, dev, 8, max(p.shape[0] for p in frames))
    with torch.no_grad():
        bev = bb([to_dev(p, dev) for p in frames]).clone()
    res = bb.results()
    assert res['overflow'] == 0
    assert np.array_equal(res['coors'].cpu().numpy(), batch8['coors4'])
    assert np.array_equal(res['num_points'].cpu().numpy(), np.concatenate(batch8['ns']))
    assert gpu_rulebook_digests(bb) == fullsize_golden()['nusc_b8']
    assert_close(bev.cpu().numpy(), batch8['expect'], 1e-4)


def test_nuscenes_batch8_bf16(dev, oracle, batch8):
    """BASELINE configs[2]: 8 distinct frames per batch on the bf16 path."""
    cfg = configs.NUSC
    frames = batch8['frames']
    bb, sd, _, ecfg, okw = build(cfg, dev, 8, max(p.shape[0] for p in frames), dtype='bf16')
    with torch.no_grad():
        bev = bb([to_dev(p, dev) for p in frames]).clone()
    res = bb.results()
    assert res['overflow'] == 0
    coors4, ns, expect = batch8['coors4'], batch8['ns'], batch8['expect']
    assert np.array_equal(res['coors'].cpu().numpy(), coors4)
    assert np.array_equal(res['num_points'].cpu().numpy(), np.concatenate(ns))
    assert gpu_rulebook_digests(bb) == fullsize_golden()['nusc_b8']
    got = bev.cpu().numpy()
    assert got.shape == expect.shape == (8, 256, 180, 180)
    assert np.abs(got - expect).max() <= 5e-2 * np.abs(expect).max()
    assert rms(got - expect) <= 2e-2 * rms(expect)
    for b in range(8):                                     # every sample, not only the aggregate
        assert rms(got[b] - expect[b]) <= 2e-2 * rms(expect[b])


def test_waymo_frame_fp32(dev, oracle):
    """BASELINE configs[4]: ~1 M points, cap of 150 000 voxels hit, HardVFE 5->64 + encoder."""
    cfg = configs.WAYMO
    frames = [synth.waymo_frame(0)]
    bb, sd, vsd, ecfg, okw = build(cfg, dev, 1, frames[0].shape[0])
    with torch.no_grad():
        bev = bb([to_dev(p, dev) for p in frames]).clone()
    res = bb.results()
    assert res['overflow'] == 0
    vs, cs, ns = voxelize_oracle(oracle, cfg, frames)
    assert cs[0].shape[0] == 150000                        # the cap triggers
    coors4 = oracle.batch_coors(cs)
    assert np.array_equal(res['coors'].cpu().numpy(), coors4)
    assert np.array_equal(res['num_points'].cpu().numpy(), ns[0])
    assert np.array_equal(res['voxels'].cpu().numpy(), vs[0])
    gold = fullsize_golden()
    assert gold['waymo_frame0_voxels'] == 150000
    assert gpu_rulebook_digests(bb) == gold['waymo_frame0']
    vcfg = cfg['pts_voxel_encoder']
    layers = [(vsd[f'vfe_layers.{i}.linear.weight'],
               {k: vsd[f'vfe_layers.{i}.norm.{k}'] for k in ('weight', 'bias', 'running_mean',
                                                             'running_var')})
              for i in range(len(vcfg['feat_channels']))]
    feats = oracle.hard_vfe(vs[0], ns[0], coors4, layers, vcfg['voxel_size'],
                            vcfg['point_cloud_range'], vcfg['norm_cfg']['eps'],
                            vcfg['with_cluster_center'], vcfg['with_voxel_center'])
    assert_close(bb.vfe_bufs[-1][:coors4.shape[0]].cpu().numpy(), feats, 1e-5, 1e-6)
    expect = oracle.sparse_encoder_forward(sd, feats, coors4, 1, ecfg['sparse_shape'], 64, **okw)
    assert expect.shape == (1, 256, 188, 188)
    assert_close(bev.cpu().numpy(), expect, 1e-4)


def test_graph_replays_equal_eager_fullsize(dev):
    """What bench.py times are graph REPLAYS: on nuScenes-shaped frames they must leave exactly the
    buffers the eager enqueue leaves (rulebooks and BEV map, bit for bit) -- for the one-graph capture
    and for the split pipeline with four objects in flight, with frames of different sizes
    following each other on the same object.  (Round 2 found a replay-only fault this way: a memset
    node inside a captured graph was not ordered with the kernel nodes around it.)"""
    import hashlib
    from sparsefusion_amd.streams import concurrent_streams
    cfg = configs.NUSC
    host = [synth.nuscenes_frame(i) for i in range(3)]
    max_pts = max(p.shape[0] for p in host)
    frames = [[to_dev(p, dev)] for p in host]

    def digest(bb):
        torch.cuda.synchronize()
        c = bb.counts.tolist()
        h = hashlib.sha256(np.asarray(c, np.int64).tobytes())
        for rb in bb.rulebooks:
            n = c[rb['dst']]
            h.update(bb.sets[rb['dst']]['indices'][:n].cpu().numpy().tobytes())
            h.update(rb['nbr'][:, :n].contiguous().cpu().numpy().tobytes())
        h.update(bb.bev.cpu().numpy().tobytes())
        return h.hexdigest()

    ref, *_ = build(cfg, dev, 1, max_pts)
    with torch.no_grad():
        want = []
        for f in frames:
            ref(f)
            assert ref.results()['overflow'] == 0
            want.append(digest(ref))
        assert len(set(want)) == len(want)
        # one graph per frame (forks inside), replayed in an order that differs from the capture order
        graphs = [ref.capture(f) for f in frames]
        for fi in (1, 0, 2, 2, 0, 1):
            graphs[fi].replay()
            assert digest(ref) == want[fi], ('capture', fi)
        del graphs
        # split pipeline, bench.py's layout
        (l0, l1, wide), _ = concurrent_streams(dev, 3)
        lanes = [LidarBackbone(cfg['pts_voxel_layer'], ref.enc, dev, max_pts, batch_size=1)
                 for _ in range(4)]
        caps = [[lane.capture_split(f, [(l0, l1)[li % 2]] * 2 + [wide]) for f in frames]
                for li, lane in enumerate(lanes)]
        for rnd in range(2):
            last = {}
            for i in range(24):
                li, fi = i % 4, (i * 2 + rnd + i // 4) % 3
                caps[li][fi].replay()
                last[li] = fi
            for li, fi in last.items():
                assert digest(lanes[li]) == want[fi], ('split', rnd, li, fi)

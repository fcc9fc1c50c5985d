"""The fused, sync-free LidarBackbone (voxelize + mean + SparseEncoder in one enqueue, counts on
the device, optional HIP graph) against the CPU oracle and against the module-by-module path."""
import numpy as np
import pytest
import torch

from helpers import (SMALL_RANGE, SMALL_SHAPE, SMALL_VOXEL, assert_close, small_points, to_dev,
                     weight_cfg)
from sparsefusion_amd.pipeline import LidarBackbone
from sparsefusion_amd.voxel_encoder import HardSimpleVFE
from sparsefusion_amd.voxelize import Voxelization
from test_gpu_conv import make_encoder

pytestmark = pytest.mark.gpu
VL = dict(max_num_points=10, voxel_size=SMALL_VOXEL, max_voxels=(3000, 6000),
          point_cloud_range=SMALL_RANGE)


def oracle_bev(oracle, sd, cfg, frames, mp, mv):
    vox, coors = [], []
    for p in frames:
        v, c, n = oracle.hard_voxelize(p, SMALL_VOXEL, SMALL_RANGE, mp, mv)
        vox.append(oracle.vfe_mean(v, n, 5))
        coors.append(c)
    feats, coors4 = np.concatenate(vox, 0), oracle.batch_coors(coors)
    bev = oracle.sparse_encoder_forward(sd, feats, coors4, len(frames), SMALL_SHAPE, 5,
                                        **weight_cfg(cfg), encoder_paddings=cfg['encoder_paddings'])
    return bev, feats, coors4


@pytest.mark.parametrize('batch', [1, 3])
def test_backbone_matches_oracle_and_modules(dev, oracle, batch):
    enc, sd, cfg = make_encoder(dev, 5, 11)
    frames = [small_points(200 + b, 3500 + 700 * b) for b in range(batch)]
    bb = LidarBackbone(VL, enc, dev, max_points=8000, batch_size=batch)
    pts = [to_dev(p, dev) for p in frames]
    with torch.no_grad():
        bev = bb(pts).clone()
    res = bb.results()
    assert res['overflow'] == 0
    expect, feats, coors4 = oracle_bev(oracle, sd, cfg, frames, 10, 6000)
    assert np.array_equal(res['coors'].cpu().numpy(), coors4)            # (b,z,y,x), batch order
    assert_close(res['voxel_features'].cpu().numpy(), feats, 1e-6, 1e-7)  # fused mean
    assert_close(bev.cpu().numpy(), expect, 1e-4)
    # module-by-module path (reference object API) gives the same tensor
    vox_layer, vfe = Voxelization(**{**VL, 'max_voxels': VL['max_voxels']}).eval(), HardSimpleVFE(5)
    with torch.no_grad():
        vs, cs, ns = zip(*[vox_layer(p) for p in pts])
        f = vfe(torch.cat(vs), torch.cat(ns), None)
        c4 = torch.cat([torch.nn.functional.pad(c, (1, 0), value=i) for i, c in enumerate(cs)])
        bev2 = enc(f, c4, batch)
    assert torch.equal(res['voxels'], torch.cat(vs))
    assert torch.equal(res['num_points'], torch.cat(ns))
    assert torch.allclose(bev, bev2, rtol=0, atol=0)   # same kernels, same order: identical


def test_graph_replay_and_reuse_across_frames(dev, oracle):
    enc, sd, cfg = make_encoder(dev, 5, 12)
    bb = LidarBackbone(VL, enc, dev, max_points=8000, batch_size=1)
    a, b = small_points(301, 5000), small_points(302, 2000)   # second frame smaller: stale state
    pa, pb = [to_dev(a, dev)], [to_dev(b, dev)]
    with torch.no_grad():
        ga, gb = bb.capture(pa), bb.capture(pb)
        for g, frame in ((ga, a), (gb, b), (ga, a)):
            g.replay()
            torch.cuda.synchronize()
            expect, _, _ = oracle_bev(oracle, sd, cfg, [frame], 10, 6000)
            assert_close(bb.bev.cpu().numpy(), expect, 1e-4)
            assert bb.results()['overflow'] == 0


def test_split_pipeline_frames_in_flight(dev):
    """capture_split: chains of linear graphs on caller-named streams, several frames in flight on
    several LidarBackbone objects without a host sync in between; every frame must come out
    bit-identical to the plain enqueue (same kernels, same order), also when an object is reused
    while later frames are still running and when a smaller frame follows a larger one."""
    from sparsefusion_amd.streams import concurrent_streams
    enc, sd, cfg = make_encoder(dev, 5, 14)
    (light0, light1, wide), nq = concurrent_streams(dev, 3)
    assert 1 <= nq <= 3
    sizes = [5000, 2000, 4200, 3100, 6000, 1500, 3600]
    pts = [[to_dev(small_points(400 + i, n), dev)] for i, n in enumerate(sizes)]
    ref = LidarBackbone(VL, enc, dev, max_points=8000, batch_size=1)
    expect = []
    with torch.no_grad():
        for p in pts:
            expect.append(ref(p).clone())
            assert ref.results()['overflow'] == 0
        lanes = [LidarBackbone(VL, enc, dev, max_points=8000, batch_size=1) for _ in range(3)]
        frames = []
        for i, p in enumerate(pts):
            light = (light0, light1)[i % 2]
            frames.append(lanes[i % 3].capture_split(p, [light, light, wide]))
        torch.cuda.synchronize()
        for rnd in range(2):
            # frames 0..6 back to back: lane i % 3 is reused every third frame without a sync
            for f in frames:
                f.replay()
            torch.cuda.synchronize()
            for i in (4, 5, 6):   # the last frame each object ran
                assert torch.equal(lanes[i % 3].bev, expect[i]), (rnd, i)
        # frame by frame with a sync: every frame, every object
        for i, f in enumerate(frames):
            f.replay()
            torch.cuda.synchronize()
            assert torch.equal(lanes[i % 3].bev, expect[i]), i
            assert lanes[i % 3].results()['overflow'] == 0
        # other cuts / one stream for everything give the same numbers
        f1 = lanes[0].capture_split(pts[0], [wide, wide, wide])
        f2 = lanes[1].capture_split(pts[0], [light0, light1, wide, light1], cut=[3, 7])
        for f, lane in ((f1, lanes[0]), (f2, lanes[1])):
            f.replay()
            torch.cuda.synchronize()
            assert torch.equal(lane.bev, expect[0])
        # a last cut == number of layers: the dense map as a segment of its own
        f3 = lanes[2].capture_split(pts[1], [light0, light0, wide, light1], cut=[11, len(lanes[2].layers)])
        assert len(f3.graphs) == 3
        f3.replay()
        torch.cuda.synchronize()
        assert torch.equal(lanes[2].bev, expect[1])
    with pytest.raises(ValueError):
        lanes[0].capture_split(pts[0], [wide, wide])
    with pytest.raises(ValueError):
        lanes[0].capture_split(pts[0], [wide, wide, wide], cut=[5, 5])


@pytest.mark.parametrize('in_flight', [3, 4])
def test_frame_pipeline_arbitrary_frames(dev, in_flight):
    """FramePipeline: frames of any size through graphs captured once per object (static point
    buffer, the rows behind the frame hold an out-of-range point); every frame equals the eager
    enqueue bit for bit -- map, coordinates, point counts -- also when a small frame follows a
    large one on the same object, from host and from device tensors."""
    from sparsefusion_amd.pipeline import FramePipeline
    enc, sd, cfg = make_encoder(dev, 5, 15)
    sizes = [5000, 2000, 4200, 3100, 6000, 1500, 3600, 5200, 800]
    host = [small_points(500 + i, n) for i, n in enumerate(sizes)]
    ref = LidarBackbone(VL, enc, dev, max_points=8000, batch_size=1)
    expect = []
    with torch.no_grad():
        for p in host:
            bev = ref([to_dev(p, dev)]).clone()
            r = ref.results()
            expect.append((bev, r['coors'].clone(), r['num_points'].clone()))
    pipe = FramePipeline(VL, enc, dev, max_points=8000, num_features=5, in_flight=in_flight)
    assert 1 <= pipe.hardware_queues <= 3
    tickets = []
    for i, p in enumerate(host):
        pts = torch.from_numpy(p) if i % 2 else to_dev(p, dev)     # host and device inputs
        tickets.append(pipe.submit(pts))
        if i >= in_flight - 1:                    # the oldest frame still held by an object
            j = i - (in_flight - 1)
            assert torch.equal(pipe.result(tickets[j]), expect[j][0]), j
            r = pipe.results(tickets[j])
            assert r['overflow'] == 0
            assert torch.equal(r['coors'], expect[j][1]) and torch.equal(r['num_points'], expect[j][2])
    for j in range(len(host) - in_flight + 1, len(host)):
        assert torch.equal(pipe.result(tickets[j]), expect[j][0]), j
    with pytest.raises(RuntimeError):
        pipe.result(tickets[0])                   # its object holds a later frame by now
    with pytest.raises(RuntimeError):
        pipe.submit(torch.zeros((9000, 5)))
    with pytest.raises(RuntimeError):
        pipe.submit(torch.zeros((10, 4)))


def test_frame_pipeline_orders_behind_the_producer(dev):
    """A device tensor handed to submit() may still be being written by work queued on the caller's
    stream (here: a long spin kernel, then the copy that fills it) and may be freed right after
    the call: the pipeline's copy has to wait for the producer and pin the memory."""
    from sparsefusion_amd.pipeline import FramePipeline
    enc, sd, cfg = make_encoder(dev, 5, 15)
    host = small_points(611, 4000)
    ref = LidarBackbone(VL, enc, dev, max_points=8000, batch_size=1)
    with torch.no_grad():
        expect = ref([to_dev(host, dev)]).clone()
    pipe = FramePipeline(VL, enc, dev, max_points=8000, num_features=5, in_flight=2)
    pipe.result(pipe.submit(to_dev(small_points(612, 3000), dev)))   # captures lane 0
    pipe.result(pipe.submit(to_dev(small_points(613, 3000), dev)))   # captures lane 1
    src = to_dev(host, dev)
    torch.cuda.synchronize()
    for _ in range(2):                                # both lanes
        pts = torch.full_like(src, 1e6)               # stale content = out-of-range points
        torch.cuda._sleep(40_000_000)                 # ~20 ms: the producer below is far from done
        pts.copy_(src)                                #   ... when submit() returns
        t = pipe.submit(pts)
        del pts                                       # the allocator may recycle it at once
        junk = torch.full_like(src, -1e6)             # (and usually hands out the same block)
        assert torch.equal(pipe.result(t), expect)
        del junk


def test_concurrent_streams(dev):
    """streams.queue_classes partitions the candidates (every stream in exactly one class, streams of
    one class really serialise: checked by the probe itself) and concurrent_streams hands out
    DISTINCT streams, from different classes while there are enough."""
    from sparsefusion_amd.streams import concurrent_streams, queue_classes
    pool = [torch.cuda.Stream(device=dev) for _ in range(6)]
    classes = queue_classes(pool)
    assert sorted(i for c in classes for i in c) == list(range(6))
    assert 1 <= len(classes) <= 6
    for n in (1, 3, 5):
        streams, found = concurrent_streams(dev, n)
        assert len(streams) == n and len({s.cuda_stream for s in streams}) == n
        assert 1 <= found <= n


def test_voxel_cap_inside_pipeline(dev, oracle):
    enc, sd, cfg = make_encoder(dev, 5, 13)
    vl = dict(VL, max_voxels=(500, 800))   # cap triggers: the reference's break semantics
    bb = LidarBackbone(vl, enc, dev, max_points=8000, batch_size=2)
    frames = [small_points(401, 4000), small_points(402, 300)]
    with torch.no_grad():
        bev = bb([to_dev(p, dev) for p in frames])
    expect, feats, coors4 = oracle_bev(oracle, sd, cfg, frames, 10, 800)
    res = bb.results()
    assert np.array_equal(res['coors'].cpu().numpy(), coors4)
    assert_close(bev.cpu().numpy(), expect, 1e-4)


def test_backbone_with_hard_vfe_waymo_style(dev, oracle):
    """Waymo-config shape of the path (configs/transfusion_waymo_voxel_L.py:109-134): voxelize
    (max 5 points) -> HardVFE 5->64 -> SparseEncoder(in_channels=64), all inside the pipeline."""
    from sparsefusion_amd import synth
    from sparsefusion_amd.voxel_encoder import HardVFE
    from helpers import torch_state_dict
    enc, sd, cfg = make_encoder(dev, 64, 31, name='waymo')
    vfe = HardVFE(in_channels=5, feat_channels=[64], voxel_size=SMALL_VOXEL,
                  point_cloud_range=SMALL_RANGE, norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01))
    vsd = synth.hard_vfe_state_dict(32, 5, [64])
    vfe.load_state_dict(torch_state_dict(vsd))
    vfe = vfe.to(dev).eval()
    vl = dict(VL, max_num_points=5, max_voxels=6000)
    frames = [small_points(700 + b, 3000 + 800 * b) for b in range(2)]
    bb = LidarBackbone(vl, enc, dev, max_points=8000, batch_size=2, voxel_encoder=vfe)
    with torch.no_grad():
        bev = bb([to_dev(p, dev) for p in frames]).clone()
    res = bb.results()
    assert res['overflow'] == 0
    vox, coors, nums = [], [], []
    for p in frames:
        v, c, n = oracle.hard_voxelize(p, SMALL_VOXEL, SMALL_RANGE, 5, 6000)
        vox.append(v), coors.append(c), nums.append(n)
    coors4 = oracle.batch_coors(coors)
    layers = [(vsd['vfe_layers.0.linear.weight'],
               {k: vsd[f'vfe_layers.0.norm.{k}'] for k in ('weight', 'bias', 'running_mean',
                                                           'running_var')})]
    feats = oracle.hard_vfe(np.concatenate(vox), np.concatenate(nums), coors4, layers, SMALL_VOXEL,
                            SMALL_RANGE, 1e-3)
    expect = oracle.sparse_encoder_forward(sd, feats, coors4, 2, SMALL_SHAPE, 64, **weight_cfg(cfg),
                                           encoder_paddings=cfg['encoder_paddings'])
    assert np.array_equal(res['coors'].cpu().numpy(), coors4)
    assert_close(bb.vfe_bufs[-1][:coors4.shape[0]].cpu().numpy(), feats, 1e-5, 1e-6)
    assert_close(bev.cpu().numpy(), expect, 1e-4)
    # module path: Voxelization -> HardVFE -> SparseEncoder gives the same tensor
    with torch.no_grad():
        f = vfe(res['voxels'], res['num_points'], res['coors'])
        bev2 = enc(f, res['coors'], 2)
    assert torch.equal(bev, bev2)

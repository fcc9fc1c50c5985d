"""SECOND / SECONDFPN on the MI355X (sf_conv2d_f32 through the C ABI) against the oracle's torch-CPU
transcription of the reference modules (mmdet3d/models/backbones/second.py:36-86,
necks/second_fpn.py:42-92).  fp32 on both sides, different summation order:
|a - b| <= 1e-4 |b| + 1e-5 max|b| (the bar of the sparse layers).  Shapes follow the reference's own
test (tests/test_models/test_necks/test_necks.py:7-45), which holds shapes only => values are
"parity unpinned" upstream and anchored on torch's CPU kernels, which the reference modules call."""
import numpy as np
import pytest
import torch

from helpers import assert_close, to_dev, torch_state_dict
from sparsefusion_amd import _lib, configs, synth
from sparsefusion_amd.second import SECOND, SECONDFPN

pytestmark = pytest.mark.gpu


def _bn(rng, c):
    return dict(weight=rng.uniform(0.5, 1.5, c).astype(np.float32),
                bias=rng.normal(0, 0.1, c).astype(np.float32),
                running_mean=rng.normal(0, 0.1, c).astype(np.float32),
                running_var=rng.uniform(0.5, 1.5, c).astype(np.float32))


def _layer(dev, x, w, bn, k, s, p, up=1, relu=True, out_channels=None, off=0, transposed=False):
    """One raw C-ABI call (no module)."""
    L = _lib.lib()
    if transposed:
        cin, cout = w.shape[:2]
        mat = np.ascontiguousarray(np.transpose(w, (2, 1, 3, 0)).reshape(up * up * cout, cin))
    else:
        cout, cin = w.shape[:2]
        mat = w
    packed = torch.empty(L.sf_conv2d_packed_bytes(cout, cin, k, up) // 4, device=dev)
    st = _lib.stream(dev)
    scale = bn['weight'] / np.sqrt(bn['running_var'] + 1e-3)
    shift = bn['bias'] - bn['running_mean'] * scale
    # device operands stay referenced until the kernel has run (a temporary's block would be reused)
    d_mat, d_x = to_dev(mat, dev), to_dev(x, dev)
    d_scale, d_shift = to_dev(scale.astype(np.float32), dev), to_dev(shift.astype(np.float32), dev)
    _lib.check(L.sf_conv2d_pack_filters_f32(d_mat.data_ptr(), cout, cin, k, up, packed.data_ptr(), st))
    B, _, H, W = x.shape
    ho, wo = ((H + 2 * p - k) // s + 1) * up, ((W + 2 * p - k) // s + 1) * up
    ctot = out_channels or cout
    out = torch.full((B, ctot, ho, wo), -7.0, device=dev)
    _lib.check(L.sf_conv2d_f32(d_x.data_ptr(), packed.data_ptr(), d_scale.data_ptr(),
                               d_shift.data_ptr(), out.data_ptr(), B, cin, H, W, cout, k, s, p, up,
                               ctot, off, int(relu), st))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize('k,s,p,cin,cout,B,H,W', [
    (3, 1, 1, 16, 64, 1, 8, 32),       # exact tiles
    (3, 1, 1, 24, 40, 2, 13, 45),      # ragged rows / columns / channels (cout, cin not multiples)
    (3, 2, 1, 16, 128, 2, 21, 37),     # the stage-entry stride-2 layer, odd sizes
    (3, 2, 1, 8, 64, 1, 90, 90),
    (1, 1, 0, 128, 256, 1, 9, 70),     # neck: conv for no stride
    (2, 2, 0, 64, 128, 2, 18, 26),     # neck: upsample_stride 0.5 (test_necks.py:24)
    (3, 1, 1, 3, 5, 1, 1, 1),          # smallest map
    (3, 1, 1, 256, 128, 1, 47, 33),    # first SECOND layer's channel counts
])
def test_conv_layer(dev, oracle, k, s, p, cin, cout, B, H, W):
    rng = np.random.default_rng(k * 1000 + cin + cout + H)
    x = rng.normal(0, 1, (B, cin, H, W)).astype(np.float32)
    w = (rng.uniform(-1, 1, (cout, cin, k, k)) * np.sqrt(3.0 / (k * k * cin))).astype(np.float32)
    bn = _bn(rng, cout)
    got = _layer(dev, x, w, bn, k, s, p)
    assert_close(got, oracle.conv2d_bn_relu(x, w, bn, s, p, 1e-3))


@pytest.mark.parametrize('up,cin,cout,B,H,W', [(2, 256, 256, 1, 12, 35), (4, 32, 24, 2, 7, 9),
                                              (1, 16, 16, 1, 5, 40)])
def test_deconv_layer_into_channel_slice(dev, oracle, up, cin, cout, B, H, W):
    rng = np.random.default_rng(up * 100 + cin)
    x = rng.normal(0, 1, (B, cin, H, W)).astype(np.float32)
    w = (rng.uniform(-1, 1, (cin, cout, up, up)) * np.sqrt(3.0 / cin)).astype(np.float32)
    bn = _bn(rng, cout)
    got = _layer(dev, x, w, bn, 1, 1, 0, up=up, out_channels=cout + 11, off=7, transposed=True)
    assert_close(got[:, 7:7 + cout], oracle.conv2d_bn_relu(x, w, bn, up, 0, 1e-3, transposed=True))
    assert np.all(got[:, :7] == -7.0) and np.all(got[:, 7 + cout:] == -7.0)   # untouched channels


def test_bad_arguments(dev):
    L = _lib.lib()
    x = torch.zeros(1, 8, 4, 4, device=dev)
    assert L.sf_conv2d_f32(x.data_ptr(), x.data_ptr(), None, None, x.data_ptr(), 1, 8, 4, 4, 8, 5, 1,
                           2, 1, 8, 0, 1, None) < 0
    assert b'not supported' in L.sf_last_error()
    assert L.sf_conv2d_f32(None, x.data_ptr(), None, None, x.data_ptr(), 1, 8, 4, 4, 8, 3, 1, 1, 1,
                           8, 0, 1, None) < 0
    assert L.sf_conv2d_f32(x.data_ptr(), x.data_ptr(), None, None, x.data_ptr(), 1, 8, 4, 4, 8, 3, 1,
                           1, 1, 8, 4, 1, None) < 0      # channel slice does not fit
    assert L.sf_conv2d_packed_bytes(0, 8, 3, 1) == 0


def _modules(dev, bcfg, ncfg, seed):
    bcfg, ncfg = dict(bcfg), dict(ncfg)
    bcfg.pop('type', None), ncfg.pop('type', None)
    bsd, nsd = synth.second_state_dict(seed, **bcfg), synth.second_fpn_state_dict(seed + 1, **ncfg)
    second, neck = SECOND(**bcfg), SECONDFPN(**ncfg)
    second.load_state_dict(torch_state_dict(bsd))
    neck.load_state_dict(torch_state_dict(nsd))
    return bcfg, ncfg, bsd, nsd, second.to(dev).eval(), neck.to(dev).eval()


def test_nuscenes_backbone_and_neck(dev, oracle):
    """The target config's SECOND + SECONDFPN (configs/sparsefusion_nusc_voxel_LC_r50.py:171-186) on
    a (1, 256, 180, 180) map with the sparsity of a BEV map (most cells zero)."""
    bcfg, ncfg, bsd, nsd, second, neck = _modules(dev, configs.NUSC['pts_backbone'],
                                                  configs.NUSC['pts_neck'], 11)
    rng = np.random.default_rng(5)
    x = (rng.normal(0, 1, (1, 256, 180, 180)) * (rng.random((1, 1, 180, 180)) < 0.4)).astype(np.float32)
    with torch.no_grad():
        feats = second(to_dev(x, dev))
        out = neck(feats)
    torch.cuda.synchronize()
    exp_feats = oracle.second_forward(bsd, x, bcfg['layer_nums'], bcfg['layer_strides'])
    exp = oracle.second_fpn_forward(nsd, exp_feats, ncfg['upsample_strides'],
                                    ncfg['use_conv_for_no_stride'])
    assert [tuple(f.shape) for f in feats] == [(1, 128, 180, 180), (1, 256, 90, 90)]
    assert isinstance(out, list) and tuple(out[0].shape) == (1, 512, 180, 180)
    for g, e in zip(feats, exp_feats):
        assert_close(g.cpu().numpy(), e)
    assert_close(out[0].cpu().numpy(), exp[0])


def test_reference_shapes_three_stage(dev, oracle):
    """tests/test_models/test_necks/test_necks.py:7-45 (shapes at its own sizes, batch 4, 512x512;
    values at batch 1, 128x128): 3-stage SECOND, the original FPN (deconv 1/2/4) and the centerpoint
    FPN (conv k=2 s=2, conv 1x1, deconv 2)."""
    bcfg = dict(in_channels=64, out_channels=[64, 128, 256], layer_nums=[3, 5, 5],
                layer_strides=[2, 2, 2])
    fpn = dict(in_channels=[64, 128, 256], upsample_strides=[1, 2, 4], out_channels=[128, 128, 128])
    cp = dict(in_channels=[64, 128, 256], out_channels=[128, 128, 128],
              upsample_strides=[0.5, 1, 2], use_conv_for_no_stride=True)
    _, fcfg, bsd, fsd, second, neck = _modules(dev, bcfg, fpn, 21)
    _, ccfg, _, csd, _, cneck = _modules(dev, bcfg, cp, 31)
    with torch.no_grad():
        big = second(torch.rand([4, 64, 512, 512], device=dev))
        assert neck(big)[0].shape == torch.Size([4, 384, 256, 256])
        assert cneck(big)[0].shape == torch.Size([4, 384, 128, 128])
        x = np.random.default_rng(9).random((1, 64, 128, 128)).astype(np.float32)
        feats = second(to_dev(x, dev))
        a, b = neck(feats)[0].cpu().numpy(), cneck(feats)[0].cpu().numpy()
    exp_feats = oracle.second_forward(bsd, x, bcfg['layer_nums'], bcfg['layer_strides'])
    assert_close(a, oracle.second_fpn_forward(fsd, exp_feats, fcfg['upsample_strides'])[0])
    assert_close(b, oracle.second_fpn_forward(csd, exp_feats, ccfg['upsample_strides'], True)[0])


def test_training_mode_bn_and_bias(dev):
    """Training-mode BatchNorm (batch statistics) and a conv bias go through the same kernel; the
    check is torch's own modules on the device (cuDNN-free: MIOpen is never called by the product,
    only by this test's reference side)."""
    torch.manual_seed(0)
    second = SECOND(in_channels=16, out_channels=[32], layer_nums=[1], layer_strides=[2],
                    conv_cfg=dict(type='Conv2d', bias=True)).to(dev)
    x = torch.randn(2, 16, 20, 28, device=dev)
    second.train()
    got = second(x)[0]
    ref_block = second.blocks[0]
    want = ref_block.cpu()(x.cpu())          # torch CPU modules = the reference's arithmetic
    second.to(dev)
    assert_close(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-3, atol_scale=1e-4)


def test_lidar_backbone_with_second_and_neck_in_one_graph(dev, oracle):
    """voxelize -> mean -> SparseEncoder -> SECOND -> SECONDFPN as ONE captured enqueue
    (detectors/sparsefusion.py:114-126) == the modules applied to the BEV map one by one (bit for
    bit: same kernels) and == the oracle chain within the fp32 bar."""
    from helpers import SMALL_RANGE, SMALL_VOXEL, small_points
    from sparsefusion_amd.pipeline import LidarBackbone
    from test_gpu_conv import make_encoder
    from test_gpu_pipeline import oracle_bev
    vl = dict(max_num_points=10, voxel_size=SMALL_VOXEL, max_voxels=(3000, 6000),
              point_cloud_range=SMALL_RANGE)
    enc, sd, cfg = make_encoder(dev, 5, 13)
    bcfg = dict(in_channels=256, out_channels=[64, 128], layer_nums=[2, 2], layer_strides=[1, 2])
    ncfg = dict(in_channels=[64, 128], out_channels=[64, 64], upsample_strides=[1, 2],
                use_conv_for_no_stride=True)
    _, _, bsd, nsd, second, neck = _modules(dev, bcfg, ncfg, 41)
    frames = [small_points(500 + b, 4000 + 500 * b) for b in range(2)]
    pts = [to_dev(p, dev) for p in frames]
    bb = LidarBackbone(vl, enc, dev, max_points=8000, batch_size=2, bev_backbone=second, neck=neck)
    with torch.no_grad():
        g = bb.capture(pts)
        g.replay()
        torch.cuda.synchronize()
        got, bev = bb.bev_out.clone(), bb.bev.clone()
        want = neck(second(bev))[0]
    assert torch.equal(got, want)
    bev_o, _, _ = oracle_bev(oracle, sd, cfg, frames, 10, 6000)
    assert_close(bev.cpu().numpy(), bev_o)
    exp = oracle.second_fpn_forward(nsd, oracle.second_forward(bsd, bev_o, bcfg['layer_nums'],
                                                               bcfg['layer_strides']),
                                    ncfg['upsample_strides'], True)[0]
    assert tuple(got.shape) == exp.shape
    assert_close(got.cpu().numpy(), exp, rtol=2e-4, atol_scale=2e-5)   # two fp32 stages chained


def test_extract_pts_feat_in_one_enqueue(dev, oracle):
    """SparseFusionDetector.extract_pts_feat (detectors/sparsefusion.py:114-131) as one enqueue:
    the neck's map with the min / max voxel-height maps appended as channels C, C+1."""
    from helpers import SMALL_RANGE, SMALL_SHAPE, SMALL_VOXEL, small_points
    from sparsefusion_amd.pipeline import LidarBackbone
    from sparsefusion_amd.voxel_heights import extract_voxel_heights
    from test_gpu_conv import make_encoder
    vl = dict(max_num_points=10, voxel_size=SMALL_VOXEL, max_voxels=(3000, 6000),
              point_cloud_range=SMALL_RANGE)
    enc, sd, cfg = make_encoder(dev, 5, 14)
    bcfg = dict(in_channels=256, out_channels=[64, 128], layer_nums=[1, 1], layer_strides=[1, 2])
    ncfg = dict(in_channels=[64, 128], out_channels=[32, 32], upsample_strides=[1, 2],
                use_conv_for_no_stride=True)
    _, _, bsd, nsd, second, neck = _modules(dev, bcfg, ncfg, 51)
    frames = [small_points(600 + b, 3000 + 900 * b) for b in range(2)]
    pts = [to_dev(p, dev) for p in frames]
    bb = LidarBackbone(vl, enc, dev, max_points=8000, batch_size=2, bev_backbone=second, neck=neck,
                       heights_out_size_factor=8)
    with torch.no_grad():
        got = bb(pts).clone()
        res = bb.results()
        want = neck(second(bb.bev))[0]
        mn, mx = extract_voxel_heights(res['voxels'].contiguous(), res['coors'].contiguous(),
                                       [SMALL_SHAPE[2], SMALL_SHAPE[1], SMALL_SHAPE[0] - 1], 8, 2)
    assert tuple(got.shape) == (2, 66, 20, 20)
    assert torch.equal(got[:, :64], want)
    assert torch.equal(got[:, 64], mn) and torch.equal(got[:, 65], mx)
    o_mn, o_mx = oracle.voxel_heights(res['voxels'].cpu().numpy(), res['coors'].cpu().numpy(), 2,
                                      [SMALL_SHAPE[2], SMALL_SHAPE[1], SMALL_SHAPE[0] - 1], 8)
    assert np.array_equal(got[:, 64].cpu().numpy(), o_mn)
    assert np.array_equal(got[:, 65].cpu().numpy(), o_mx)

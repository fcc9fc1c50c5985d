"""SparseInverseConv3d and SparseUNet (mmdet3d/ops/spconv/conv.py:147-152,359-374,
mmdet3d/models/middle_encoders/sparse_unet.py:8-292; SURVEY 8f row 4, last item).

The reference tests hold construction / shape checks only
(tests/test_models/test_common_modules/test_sparse_unet.py:8-49: module types, channel counts,
spatial_features [2, 256, 5, 5], seg_features [4, 16] on a 4-voxel input) => values "parity
unpinned" upstream.  The oracle's inverse convolution is pinned by dense equivalence: with pairs
in = out*s - p + k the inverse convolution is torch.nn.functional.conv_transpose3d of the densified
coarse tensor, read at the fine active set.  The HIP path is compared with the oracle wiring."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import SMALL_RANGE, SMALL_SHAPE, SMALL_VOXEL, assert_close, small_points, to_dev, \
    torch_state_dict
from sparsefusion_amd import synth


@pytest.mark.parametrize('pad,shape', [(1, [13, 21, 24]), (0, [11, 19, 23]), ((0, 1, 1), [11, 20, 24])])
def test_oracle_inverse_conv_is_dense_transposed_conv(oracle, pad, shape):
    rng = np.random.default_rng(3)
    batch, cin, cout, n = 2, 6, 5, 400
    coords = np.unique(np.stack([rng.integers(0, batch, n), rng.integers(0, shape[0], n),
                                 rng.integers(0, shape[1], n), rng.integers(0, shape[2], n)], 1),
                       axis=0).astype(np.int32)
    coords = coords[rng.permutation(coords.shape[0])]
    outids, pairs, num, oshape = oracle.get_indice_pairs(coords, batch, shape, 3, 2, pad, 1, False)
    coarse = rng.normal(0, 1, (outids.shape[0], cin)).astype(np.float32)
    w = rng.normal(0, 0.3, (3, 3, 3, cin, cout)).astype(np.float32)        # spconv-1 layout
    y = oracle.indice_inverse_conv(coarse, w.reshape(27, cin, cout), pairs, num, coords.shape[0])
    dense = np.zeros((batch, cin, *oshape), np.float64)
    dense[outids[:, 0], :, outids[:, 1], outids[:, 2], outids[:, 3]] = coarse
    p3 = list(pad) if isinstance(pad, tuple) else [pad] * 3
    opad = [shape[d] - ((oshape[d] - 1) * 2 - 2 * p3[d] + 3) for d in range(3)]
    wt = torch.from_numpy(np.transpose(w, (3, 4, 0, 1, 2)).copy()).double()   # (ci, co, kz, ky, kx)
    ref = F.conv_transpose3d(torch.from_numpy(dense), wt, stride=2, padding=tuple(p3),
                             output_padding=tuple(opad)).numpy()
    assert list(ref.shape[2:]) == shape
    np.testing.assert_allclose(y, ref[coords[:, 0], :, coords[:, 1], coords[:, 2], coords[:, 3]],
                               rtol=1e-4, atol=1e-4)


def test_construction_matches_reference():
    """tests/test_models/test_common_modules/test_sparse_unet.py:8-37."""
    from sparsefusion_amd import spconv
    from sparsefusion_amd.sparse_block import SparseBasicBlock
    from sparsefusion_amd.sparse_unet import SparseUNet
    unet = SparseUNet(in_channels=4, sparse_shape=[41, 1600, 1408])
    assert len(unet.encoder_layers) == 4
    assert unet.encoder_layers.encoder_layer1[0][0].in_channels == 16
    assert unet.encoder_layers.encoder_layer1[0][0].out_channels == 16
    assert isinstance(unet.encoder_layers.encoder_layer1[0][0], spconv.SubMConv3d)
    assert isinstance(unet.encoder_layers.encoder_layer1[0][1], torch.nn.BatchNorm1d)
    assert isinstance(unet.encoder_layers.encoder_layer1[0][2], torch.nn.ReLU)
    assert unet.encoder_layers.encoder_layer4[0][0].in_channels == 64
    assert unet.encoder_layers.encoder_layer4[0][0].out_channels == 64
    assert isinstance(unet.encoder_layers.encoder_layer4[0][0], spconv.SparseConv3d)
    assert isinstance(unet.encoder_layers.encoder_layer4[2][0], spconv.SubMConv3d)
    assert isinstance(unet.lateral_layer1, SparseBasicBlock)
    assert isinstance(unet.merge_layer1[0], spconv.SubMConv3d)
    assert isinstance(unet.upsample_layer1[0], spconv.SubMConv3d)
    assert isinstance(unet.upsample_layer2[0], spconv.SparseInverseConv3d)
    assert unet.upsample_layer2[0].indice_key == 'spconv2' and unet.upsample_layer2[0].inverse
    sd = synth.sparse_unet_state_dict(1, 4)
    assert set(sd) == set(unet.state_dict())


def _unet(dev, seed, in_channels):
    from sparsefusion_amd.sparse_unet import SparseUNet
    unet = SparseUNet(in_channels=in_channels, sparse_shape=SMALL_SHAPE)
    sd = synth.sparse_unet_state_dict(seed, in_channels)
    unet.load_state_dict(torch_state_dict(sd))
    return unet.to(dev).eval(), sd


@pytest.mark.gpu
def test_reference_shape_test(dev):
    """test_sparse_unet.py:38-49: 4 voxels, batch 2 -> spatial [2, 256, 5, 5], seg [4, 16]."""
    from sparsefusion_amd.sparse_unet import SparseUNet
    unet = SparseUNet(in_channels=4, sparse_shape=[41, 40, 40]).to(dev).eval()
    voxel_features = torch.tensor([[6.56126, 0.9648336, -1.7339306, 0.315],
                                   [6.8162713, -2.480431, -1.3616394, 0.36],
                                   [11.643568, -4.744306, -1.3580885, 0.16],
                                   [23.482342, 6.5036807, 0.5806964, 0.35]], device=dev)
    coordinates = torch.tensor([[0, 12, 19, 31], [0, 16, 20, 12], [1, 10, 15, 33], [1, 2, 33, 17]],
                               dtype=torch.int32, device=dev)
    with torch.no_grad():
        ret = unet(voxel_features, coordinates, 2)
    assert ret['spatial_features'].shape == torch.Size([2, 256, 5, 5])
    assert ret['seg_features'].shape == torch.Size([4, 16])


@pytest.mark.gpu
@pytest.mark.parametrize('batch', [1, 2])
def test_sparse_unet_matches_oracle(dev, oracle, batch):
    unet, sd = _unet(dev, 7, 4)
    frames = [small_points(800 + b, 3000 + 1000 * b, 4) for b in range(batch)]
    vox = [oracle.hard_voxelize(p, SMALL_VOXEL, SMALL_RANGE, 5, 4000) for p in frames]
    feats = np.concatenate([oracle.vfe_mean(v, n, 4) for v, _, n in vox])
    coors = oracle.batch_coors([c for _, c, _ in vox])
    with torch.no_grad():
        ret = unet(to_dev(feats, dev), to_dev(coors, dev), batch)
    torch.cuda.synchronize()
    spatial, seg, seg_idx = oracle.sparse_unet_forward(sd, feats, coors, batch, SMALL_SHAPE, 4)
    assert np.array_equal(seg_idx, coors)        # the decoder ends on the input index set, in order
    assert tuple(ret['spatial_features'].shape) == spatial.shape
    assert_close(ret['spatial_features'].cpu().numpy(), spatial, 1e-4)
    assert tuple(ret['seg_features'].shape) == seg.shape == (coors.shape[0], 16)
    # 20 fp32 layers chained through an additive decoder: twice the single-encoder bar
    assert_close(ret['seg_features'].cpu().numpy(), seg, rtol=2e-4, atol_scale=2e-5)


@pytest.mark.gpu
def test_inverse_conv_layer_matches_oracle(dev, oracle):
    """One SparseConv3d / SparseInverseConv3d couple through the module API (shared indice_key)."""
    from sparsefusion_amd import spconv
    rng = np.random.default_rng(11)
    p = small_points(900, 5000, 4)
    _, c, _ = oracle.hard_voxelize(p, SMALL_VOXEL, SMALL_RANGE, 5, 6000)
    coors = oracle.batch_coors([c])
    feats = rng.normal(0, 1, (coors.shape[0], 16)).astype(np.float32)
    down = spconv.SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False, indice_key='d').to(dev)
    up = spconv.SparseInverseConv3d(32, 16, 3, indice_key='d', bias=False).to(dev)
    x = spconv.SparseConvTensor(to_dev(feats, dev), to_dev(coors, dev), SMALL_SHAPE, 1)
    with torch.no_grad():
        mid = down(x)
        y = up(mid)
    torch.cuda.synchronize()
    assert y.spatial_shape == SMALL_SHAPE and torch.equal(y.indices, x.indices)
    outids, pairs, num, _ = oracle.get_indice_pairs(coors, 1, SMALL_SHAPE, 3, 2, 1, 1, False)
    wd = oracle.weight_to_kio(down.weight.detach().cpu().numpy(), 3, 16, 32)
    wu = oracle.weight_to_kio(up.weight.detach().cpu().numpy(), 3, 32, 16)
    m = oracle.indice_conv(feats, wd, pairs, num, outids.shape[0], False)
    # row-match the strided output (the HIP rulebook keeps the CPU reference's first-touch order)
    assert np.array_equal(mid.indices.cpu().numpy(), outids)
    assert_close(mid.features.cpu().numpy(), m, 1e-4)
    e = oracle.indice_inverse_conv(m, wu, pairs, num, coors.shape[0])
    assert_close(y.features.cpu().numpy(), e, 1e-4)

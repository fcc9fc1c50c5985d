"""`LidarBackbone`: the whole hot path as ONE sync-free enqueue on one HIP stream.

    points (list of (N_i,F) f32 per sample)
      -> sf_hard_voxelize (+ fused HardSimpleVFE mean, batch index, row append)   per sample
      -> coordinate phase: hash tables + neighbour tables of every layer geometry
      -> feature phase: 21 fused conv+BN(+residual)+ReLU kernels
      -> sf_dense_bev                                  -> (B, C*D, H, W) f32

It is what detectors/sparsefusion.py:114-122 does through Voxelization / HardSimpleVFE /
SparseEncoder (voxelize -> pts_voxel_encoder -> pts_middle_encoder), with the data-dependent
row counts left in device memory: every buffer is sized once from host-known upper bounds, no
kernel argument depends on a count, so nothing waits on the host and the whole frame can be
captured in a HIP graph (`capture()`), as a launch-bound inner loop should be on MI355X.

The modules in voxelize.py / sparse_encoder.py give the same numbers through the reference's
object API; this class takes its layer list and weights FROM a SparseEncoder module.
"""
import torch
from torch import nn

from . import _lib, spconv
from .sparse_block import SparseBasicBlock
from .sparse_encoder import SparseEncoder


class _Layer:
    __slots__ = ('conv', 'bn', 'relu', 'residual_from', 'rb', 'cin', 'cout', 'packed', 'scale',
                 'shift', 'in_buf', 'out_buf', 'bf16', 'out_dtype', 'ordered')


class _SplitFrame:
    """One batch captured as a chain of linear graphs (LidarBackbone.capture_split), segment i on
    streams[i].  replay() launches them in order, each waiting for the event of the one before;
    the first waits until the last segment of this object's PREVIOUS batch is done (its buffers).

    The events are plain torch events (device-scope release at the record): an event created with
    hipEventDisableSystemFence carries NO cache release, and a consumer on another hardware queue
    -- other XCDs, other L2s -- then reads stale lines of the previous frame: measured as an
    illegal address in the wide layers (a plan half old, half new) on the small test frames."""

    def __init__(self, owner, graphs, streams):
        self.owner, self.graphs, self.streams = owner, graphs, streams
        self.events = [torch.cuda.Event() for _ in graphs]

    def replay(self, timed=None):
        """timed: a list; a (start, end) pair of timing events around the LAST segment is appended
        (bench.py: duration of the wide layers inside the running pipeline)."""
        o = self.owner
        prev = o._last_segment_done
        last = len(self.graphs) - 1
        for i, (g, st, ev) in enumerate(zip(self.graphs, self.streams, self.events)):
            if prev is not None:
                st.wait_event(prev)
            with torch.cuda.stream(st):
                if timed is not None and i == last:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    g.replay()
                    e1.record(st)
                    timed.append((e0, e1))
                else:
                    g.replay()
                ev.record(st)
            prev = ev
        o._last_segment_done = prev


def _flatten_encoder(enc):
    """SparseEncoder module tree -> [(conv, bn, relu, identity_role)] in forward
    order (sparse_encoder.py:112-121, sparse_block.py:116-139).  Only the post-activation
    ('conv','norm','act') order with eval-mode BatchNorm1d is fusable."""
    out = []

    def convmodule(seq):
        mods = list(seq)
        if not (len(mods) == 3 and isinstance(mods[0], spconv.SparseConvolution) and
                isinstance(mods[1], nn.BatchNorm1d) and isinstance(mods[2], nn.ReLU)):
            raise NotImplementedError('LidarBackbone fuses conv->BN->ReLU modules only')
        out.append((mods[0], mods[1], True, None))

    convmodule(enc.conv_input)
    for stage in enc.encoder_layers:
        for blk in stage:
            if isinstance(blk, SparseBasicBlock):
                out.append((blk.conv1, blk.norm1, True, 'save'))    # block input = identity
                out.append((blk.conv2, blk.norm2, True, 'add'))     # + identity before ReLU
            else:
                convmodule(blk)
    convmodule(enc.conv_out)
    return out


class LidarBackbone:

    def __init__(self, voxel_layer_cfg, encoder, device, max_points, batch_size=1,
                 materialize_voxels=True, training=False, overlap=True, dtype='f32',
                 voxel_encoder=None, bev_backbone=None, neck=None,
                 heights_out_size_factor=None, strided_growth=None, fused_front=True):
        """dtype 'f32': every layer on the exact-fp32 kernels.  'bf16' (BASELINE configs[2]): the
        first layer stays fp32 (raw metric coordinates need more than 8 significant bits), its
        output is cast to bf16, every later layer runs on the bf16 matrix cores with fp32
        accumulation and the last one writes fp32 for the dense map."""
        assert isinstance(encoder, SparseEncoder)
        assert dtype in ('f32', 'bf16')
        self.dtype = dtype
        # strided_growth: bound on (output rows / input rows) of a strided convolution used to size
        # buffers and hash tables.  None = the geometric worst case (8 for 3x3x3 / stride 2: always
        # safe).  LiDAR frames stay far below it (1.3, 0.6, 0.33 on the nuScenes-shaped frames); a
        # smaller value shrinks the hash tables (fewer cache misses per probe, less to clear) and
        # is guarded by the overflow flag: results() reports it and nothing is written out of bounds.
        self.strided_growth = None if strided_growth is None else float(strided_growth)
        # bev_backbone / neck: optional SECOND / SECONDFPN modules (second.py) run on the dense map
        # in the same enqueue (detectors/sparsefusion.py:123-126); enqueue then returns what they
        # return (the neck's single map, or the backbone's tuple) instead of the BEV map
        self.bev_backbone, self.neck = bev_backbone, neck
        if neck is not None and bev_backbone is None:
            raise ValueError('a neck needs a bev_backbone')
        self.bev_out = None
        # heights_out_size_factor: also run SparseFusion's extract_voxel_heights on the voxels of
        # the same voxelization (detectors/sparsefusion.py:73-112,159) and append the min / max
        # height maps as two more channels of the neck's map (:127-130): the enqueue is then the
        # whole of SparseFusionDetector.extract_pts_feat (:114-131)
        self.osf = None if heights_out_size_factor is None else int(heights_out_size_factor)
        if self.osf is not None:
            if neck is None:
                raise ValueError('the height channels are appended to the neck output')
            materialize_voxels = True
        self.min_height = self.max_height = None
        # voxel_encoder: None = the fused HardSimpleVFE mean (nuScenes config); a HardVFE module
        # (Waymo config, voxel_encoder.py:283-450) runs its VFE layers on the materialised voxels
        self.vfe = voxel_encoder
        if voxel_encoder is not None:
            from .voxel_encoder import HardVFE
            if not isinstance(voxel_encoder, HardVFE) or voxel_encoder._with_distance:
                raise NotImplementedError('LidarBackbone takes a HardVFE (without with_distance)')
            materialize_voxels = True
        L = self.L = _lib.lib()
        self.dev = torch.device(device)
        self.B = int(batch_size)
        self.max_points_in = int(max_points)
        cfg = voxel_layer_cfg
        self.voxel_size = list(cfg['voxel_size'])
        self.pc_range = list(cfg['point_cloud_range'])
        self.mp = int(cfg['max_num_points'])
        mv = cfg['max_voxels']
        mv = mv if isinstance(mv, (tuple, list)) else (mv, mv)
        self.mv = int(mv[0] if training else mv[1])
        self.materialize = materialize_voxels
        self.overlap = bool(overlap)      # coordinate phase on a second stream
        # fused_front: wherever the coordinate phase runs on ONE stream (capture_split, overlap=False)
        # it is enqueued by one sf_lidar_front call (fewer launches, same buffers) instead of the
        # stand-alone entry points
        self.fused_front = bool(fused_front)
        self._front_desc = None
        self.side_stream = torch.cuda.Stream(device=self.dev) if self.overlap else None
        self._last_segment_done = None   # capture_split: event at the end of this object's last batch
        self.enc = encoder
        self.in_channels = encoder.in_channels
        self.F = None  # point feature count, fixed at first call
        dev = self.dev

        def i32(*shape):
            return torch.empty(shape, dtype=torch.int32, device=dev)

        def f32(*shape):
            return torch.empty(shape, dtype=torch.float32, device=dev)

        # ---- geometry plan: coordinate sets and rulebooks -------------------------------
        flat = _flatten_encoder(encoder)
        shape = [int(s) for s in encoder.sparse_shape]
        n0 = self.B * self.mv
        self.sets = [dict(shape=shape, bound=n0)]          # coordinate set 0 = voxels
        self.rulebooks = []                                # dicts, deduplicated
        self.layers = []
        cur = 0
        rb_index = {}
        for conv, bn, relu, res in flat:
            ks, dl = conv.kernel_size, conv.dilation
            if any(d != 1 for d in dl):
                raise NotImplementedError('dilation != 1')
            if conv.subm:
                st, pd = [1, 1, 1], [k // 2 for k in ks]
            else:
                st, pd = conv.stride, conv.padding
            key = (cur, tuple(ks), tuple(st), tuple(pd), bool(conv.subm))
            if key not in rb_index:
                if conv.subm:
                    dst = cur
                else:
                    s_in = self.sets[cur]
                    oshape = spconv.get_conv_output_size(s_in['shape'], ks, st, pd, dl)
                    reach = 1
                    for k, s in zip(ks, st):
                        reach *= (k + s - 1) // s
                    vol = self.B * oshape[0] * oshape[1] * oshape[2]
                    if self.strided_growth is not None:
                        reach = min(reach, self.strided_growth)
                    self.sets.append(dict(shape=oshape,
                                          bound=max(min(int(s_in['bound'] * reach), vol), 1)))
                    dst = len(self.sets) - 1
                rb_index[key] = len(self.rulebooks)
                self.rulebooks.append(dict(src=cur, dst=dst, ks=list(ks), st=list(st), pd=list(pd),
                                           subm=bool(conv.subm), K=ks[0] * ks[1] * ks[2]))
            rb = self.rulebooks[rb_index[key]]
            lay = _Layer()
            lay.conv, lay.bn, lay.relu, lay.residual_from = conv, bn, relu, res
            lay.rb, lay.cin, lay.cout = rb, conv.in_channels, conv.out_channels
            lay.bf16 = dtype == 'bf16' and len(self.layers) > 0
            lay.out_dtype = torch.bfloat16 if lay.bf16 else torch.float32
            self.layers.append(lay)
            cur = rb['dst']
        if dtype == 'bf16' and len(self.layers) > 1:
            self.layers[-1].out_dtype = torch.float32
        self.out_set = cur
        self.out_channels = self.layers[-1].cout

        # ---- device buffers sized from the bounds ---------------------------------------
        self.counts = torch.zeros(len(self.sets) + 1, dtype=torch.int32, device=dev)  # + overflow
        for s in self.sets:
            b = s['bound']
            s['indices'] = i32(b, 4)
            s['cap'] = L.sf_hash_capacity(b)
            s['keys'] = i32(s['cap'])
            s['vals'] = i32(s['cap'])
        for rb in self.rulebooks:
            nb = spconv.nbr_stride(self.sets[rb['dst']]['bound'])
            rb['nbr'] = i32(rb['K'], nb)
            rb['stride'] = nb
        # row order by neighbour mask for the rulebooks whose fp32 layers run on the ordered kernel
        for lay in self.layers:
            lay.ordered = (not lay.bf16) and spconv.use_ordered(L.sf_conv_padded_cin(lay.cin), lay.cout, lay.rb['subm'])
            if lay.ordered and 'order' not in lay.rb:
                nb_rows = self.sets[lay.rb['dst']]['bound']
                lay.rb['order'] = i32(L.sf_row_order_size(nb_rows))
                lay.rb['order_ws_bytes'] = L.sf_row_order_workspace(nb_rows)
                lay.rb['order_ws'] = torch.empty(lay.rb['order_ws_bytes'], dtype=torch.uint8,
                                                 device=dev)
        ws = 0
        for rb in self.rulebooks:
            if not rb['subm']:
                ws = max(ws, L.sf_strided_rulebook_workspace(
                    self.sets[rb['src']]['bound'], self.sets[rb['dst']]['cap'], _lib.i3(rb['ks']),
                    _lib.i3(rb['st'])))
        self.rb_ws = torch.empty(max(ws, 256), dtype=torch.uint8, device=dev)
        self.rb_ws_bytes = max(ws, 256)
        vws = L.sf_hard_voxelize_workspace(self.max_points_in, self.mp)
        vws = (vws + 255) & ~255
        self.vox_ws = torch.empty(vws * self.B, dtype=torch.uint8, device=dev)   # a slice per sample
        self.vox_ws_bytes = vws
        for rb in self.rulebooks:   # first-touch words of a strided rulebook (sf_lidar_front)
            if not rb['subm']:
                rb['first'] = i32(self.sets[rb['dst']]['cap'])
        self.cin_pad = L.sf_conv_padded_cin(self.in_channels)
        self.mean = f32(n0, self.cin_pad)                  # fused VFE output, zero padded columns
        self.voxels = None                                  # allocated once F is known
        self.num_points = i32(n0)
        # feature buffers: three per coordinate set are enough for conv / block ping-pong
        self.fbufs = {}
        for lay in self.layers:
            d = lay.rb['dst']
            self.fbufs.setdefault((d, lay.cout, lay.out_dtype),
                                  [torch.empty((self.sets[d]['bound'], lay.cout),
                                               dtype=lay.out_dtype, device=dev) for _ in range(3)])
        self.vfe_bufs = []
        if self.vfe is not None:
            for lay in self.vfe.vfe_layers:
                if lay.norm.training:
                    raise RuntimeError('LidarBackbone needs eval-mode BatchNorm in the voxel encoder')
                co = lay.linear.out_features
                shape = (n0, self.mp, 2 * co) if lay.cat_max else (n0, co)
                self.vfe_bufs.append(f32(*shape))
            if self.vfe_bufs[-1].shape[1] != self.cin_pad:
                raise RuntimeError('voxel encoder output channels must equal the (padded) encoder '
                                   'input channels')
        self.cast0 = None
        if dtype == 'bf16' and len(self.layers) > 1:     # bf16 copy of the first layer's output
            l0 = self.layers[0]
            self.cast0 = torch.empty((self.sets[l0.rb['dst']]['bound'],
                                      L.sf_conv_padded_cin_bf16(l0.cout)), dtype=torch.bfloat16,
                                     device=dev)
        D, H, W = self.sets[self.out_set]['shape']
        self.bev = f32(self.B, self.out_channels * D, H, W)
        self.refresh_weights()

    # ------------------------------------------------------------------
    def refresh_weights(self):
        """(Re)pack filters and fold BatchNorm from the SparseEncoder module's current tensors."""
        for lay in self.layers:
            if lay.bn.training:
                raise RuntimeError('LidarBackbone needs eval-mode BatchNorm (call encoder.eval())')
            lay.packed = lay.conv.packed_filters(torch.bfloat16 if lay.bf16 else torch.float32)
            lay.scale, lay.shift = spconv.fold_bn(lay.bn)
            if lay.conv.bias is not None:
                lay.shift = (lay.shift + lay.conv.bias.detach().float() * lay.scale).contiguous()
        self.vfe_params = []
        if self.vfe is not None:
            for vl in self.vfe.vfe_layers:
                sc, sh = spconv.fold_bn(vl.norm)
                self.vfe_params.append((vl.linear.weight.detach().float().contiguous(), sc, sh))

    def memory_bytes(self):
        tot = 0
        for t in self._tensors():
            tot += t.numel() * t.element_size()
        return tot

    def _tensors(self):
        for s in self.sets:
            yield from (s['indices'], s['keys'], s['vals'])
        for rb in self.rulebooks:
            yield rb['nbr']
            if 'first' in rb:
                yield rb['first']
        for bufs in self.fbufs.values():
            yield from bufs
        yield from (self.rb_ws, self.vox_ws, self.mean, self.num_points, self.bev)
        yield from self.vfe_bufs
        if self.cast0 is not None:
            yield self.cast0
        if self.voxels is not None:
            yield self.voxels

    # ------------------------------------------------------------------
    def enqueue(self, points_list, voxel_features=None):
        """Enqueue one batch on the current stream; returns the (static) BEV tensor.  No host
        synchronisation, no allocation after the first call: safe under graph capture.

        voxel_features: optional (>= B*max_voxels, in_channels) f32 tensor used as the encoder
        input instead of the fused per-voxel mean (configs whose voxel encoder is a learned VFE,
        e.g. the Waymo config's 64-channel HardVFE, which is not part of this path)."""
        dev = self.dev
        self._dense_marked = False
        if self.fused_front and not (self.overlap and len(self.rulebooks) > 1):
            self._front_fused(points_list, voxel_features)
            self._feature_input(voxel_features)
            self._features(0, len(self.layers), {})
            return self._dense_and_after()
        self._front(points_list, voxel_features)
        # ---- coordinate phase: on its own stream, forked after the voxelizer ---------------
        # The hash / neighbour-table kernels are latency bound and need no LDS: they run in the
        # shadow of the MFMA-bound convolutions of the earlier stages.  Each rulebook records an
        # event; the feature phase waits for it before the first layer that uses the rulebook
        # (under graph capture these become fork / join edges of the graph).
        main = torch.cuda.current_stream(dev)
        fork = self.overlap and len(self.rulebooks) > 1
        side = self.side_stream if fork else main
        if fork:
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
        rb_ready = {}
        for ri in range(1, len(self.rulebooks)):
            self._rulebook(ri, side.cuda_stream)
            if fork:
                e = torch.cuda.Event()
                e.record(side)
                rb_ready[ri] = e
        self._feature_input(voxel_features)
        self._features(0, len(self.layers), rb_ready)
        return self._dense_and_after()

    def _front(self, points_list, voxel_features):
        """Voxelizer (+ mean, + batch index), hash of the voxel coordinates and the FIRST rulebook
        (it feeds the very first layer), all on the current stream."""
        L, dev = self.L, self.dev
        st = _lib.stream(dev)
        chk = _lib.check
        if len(points_list) != self.B:
            raise RuntimeError(f'expected {self.B} samples, got {len(points_list)}')
        if self.F is None:
            self.F = int(points_list[0].shape[1])
            if self.materialize:
                self.voxels = torch.empty((self.B * self.mv, self.mp, self.F), dtype=torch.float32,
                                          device=dev)
        cnt = self.counts
        c0 = cnt.data_ptr()
        ovf = c0 + 4 * len(self.sets)
        self.counts.zero_()
        s0 = self.sets[0]
        if voxel_features is None and self.vfe is None and self.in_channels > self.F:
            raise RuntimeError(f'encoder wants {self.in_channels} input channels but points have '
                               f'{self.F} features: pass voxel_features')
        nf = min(self.in_channels, self.F)
        # samples append behind each other
        for b, pts in enumerate(points_list):
            if pts.shape[0] > self.max_points_in or pts.shape[1] != self.F:
                raise RuntimeError('point cloud larger than max_points or feature count changed')
            _lib.require(pts, torch.float32, 'points')
            chk(L.sf_hard_voxelize(
                _lib.ptr(pts), pts.shape[0], self.F, _lib.f3(self.voxel_size),
                _lib.f6(self.pc_range), self.mp, self.mv,
                self.voxels.data_ptr() if self.materialize else None, s0['indices'].data_ptr(), 4,
                b, self.num_points.data_ptr(),
                self.mean.data_ptr() if self.vfe is None else None, nf, self.cin_pad, c0,
                s0['bound'], ovf, None, self.vox_ws.data_ptr(), self.vox_ws_bytes, st))
        chk(L.sf_hash_build(s0['indices'].data_ptr(), c0, s0['bound'], _lib.i3(s0['shape']),
                            self.B, s0['keys'].data_ptr(), s0['vals'].data_ptr(), s0['cap'], ovf,
                            st))
        self._rulebook(0, st)

    def _front_fused(self, points_list, voxel_features):
        """The whole coordinate phase (voxelizer, hashes, every rulebook, plans) by ONE call of
        sf_lidar_front on the current stream."""
        import ctypes as C
        L, dev = self.L, self.dev
        if len(points_list) != self.B:
            raise RuntimeError(f'expected {self.B} samples, got {len(points_list)}')
        if self.F is None:
            self.F = int(points_list[0].shape[1])
            if self.materialize:
                self.voxels = torch.empty((self.B * self.mv, self.mp, self.F), dtype=torch.float32,
                                          device=dev)
        if voxel_features is None and self.vfe is None and self.in_channels > self.F:
            raise RuntimeError(f'encoder wants {self.in_channels} input channels but points have '
                               f'{self.F} features: pass voxel_features')
        for pts in points_list:
            if pts.shape[0] > self.max_points_in or pts.shape[1] != self.F:
                raise RuntimeError('point cloud larger than max_points or feature count changed')
            _lib.require(pts, torch.float32, 'points')
        if self._front_desc is None:
            sets = (_lib.SfCoordSet * len(self.sets))()
            for cs, s in zip(sets, self.sets):
                cs.shape[:] = [int(v) for v in s['shape']]
                cs.bound, cs.capacity = s['bound'], s['cap']
                cs.indices, cs.keys, cs.vals = s['indices'].data_ptr(), s['keys'].data_ptr(), s['vals'].data_ptr()
            rbs = (_lib.SfRulebook * len(self.rulebooks))()
            ws = 0
            for cr, rb in zip(rbs, self.rulebooks):
                cr.src, cr.dst, cr.subm = rb['src'], rb['dst'], 1 if rb['subm'] else 0
                cr.ksize[:], cr.stride[:], cr.padding[:] = rb['ks'], rb['st'], rb['pd']
                cr.nbr, cr.nbr_stride = rb['nbr'].data_ptr(), rb['stride']
                if 'order' in rb:
                    cr.plan, cr.plan_ws = rb['order'].data_ptr(), rb['order_ws'].data_ptr()
                    cr.plan_ws_bytes = rb['order_ws_bytes']
                if 'first' in rb:
                    cr.first = rb['first'].data_ptr()
            d = _lib.SfFrontDesc()
            d.batch = self.B
            d.points = (C.c_void_p * self.B)()
            d.num_points = (C.c_int * self.B)()
            d.num_features = self.F
            d.voxel_size[:] = [float(v) for v in self.voxel_size]
            d.coors_range[:] = [float(v) for v in self.pc_range]
            d.max_points, d.max_voxels = self.mp, self.mv
            d.voxels = self.voxels.data_ptr() if self.materialize else None
            d.num_points_per_voxel = self.num_points.data_ptr()
            d.mean = self.mean.data_ptr() if self.vfe is None else None
            d.mean_features, d.mean_stride = min(self.in_channels, self.F), self.cin_pad
            d.vox_workspace, d.vox_workspace_bytes = self.vox_ws.data_ptr(), self.vox_ws_bytes
            d.n_sets, d.sets, d.counts_dev = len(self.sets), sets, self.counts.data_ptr()
            d.n_rulebooks, d.rulebooks = len(self.rulebooks), rbs
            d.rb_workspace, d.rb_workspace_bytes = self.rb_ws.data_ptr(), self.rb_ws_bytes
            self._front_desc = (d, sets, rbs, d.points, d.num_points)
        d = self._front_desc[0]
        pts_arr = (C.c_void_p * self.B)(*[_lib.ptr(p) for p in points_list])
        n_arr = (C.c_int * self.B)(*[int(p.shape[0]) for p in points_list])
        d.points, d.num_points = pts_arr, n_arr
        _lib.check(L.sf_lidar_front(C.byref(d), _lib.stream(dev)))

    def _rulebook(self, ri, ss):
        """Rulebook `ri` (and its mask order, where a layer uses one) on the raw stream `ss`."""
        L, chk = self.L, _lib.check
        rb = self.rulebooks[ri]
        c0 = self.counts.data_ptr()
        ovf = c0 + 4 * len(self.sets)
        src, dst = self.sets[rb['src']], self.sets[rb['dst']]
        n_src, n_dst = c0 + 4 * rb['src'], c0 + 4 * rb['dst']
        if not rb['subm']:
            # output set + neighbour table from the input side (one probe per pair)
            chk(L.sf_strided_rulebook(
                src['indices'].data_ptr(), n_src, src['bound'], self.B, _lib.i3(dst['shape']),
                _lib.i3(rb['ks']), _lib.i3(rb['st']), _lib.i3(rb['pd']), dst['keys'].data_ptr(),
                dst['vals'].data_ptr(), dst['cap'], dst['indices'].data_ptr(), dst['bound'],
                n_dst, ovf, rb['nbr'].data_ptr(), rb['stride'], self.rb_ws.data_ptr(),
                self.rb_ws_bytes, ss))
        else:
            chk(L.sf_neighbor_table(
                dst['indices'].data_ptr(), n_dst, dst['bound'], self.B, _lib.i3(src['shape']),
                _lib.i3(rb['ks']), _lib.i3(rb['st']), _lib.i3(rb['pd']), src['keys'].data_ptr(),
                src['vals'].data_ptr(), src['cap'], 1 if rb['subm'] else 0, rb['nbr'].data_ptr(),
                rb['stride'], ss))
        self._row_order(rb, n_dst, dst['bound'], ss)

    def _feature_input(self, voxel_features):
        """Encoder input of this batch: the fused mean, the HardVFE layers' output, or the caller's
        voxel_features.  Leaves the running (x, channels, block input) state for _features."""
        L, dev = self.L, self.dev
        st = _lib.stream(dev)
        chk = _lib.check
        c0 = self.counts.data_ptr()
        s0 = self.sets[0]
        x, x_ch = self.mean, self.cin_pad
        if self.vfe is not None and voxel_features is None:
            v = self.vfe
            deco = (1 if v._with_cluster_center else 0) | (2 if v._with_voxel_center else 0) | 8
            cur, cur_c = self.voxels, self.F
            from .voxel_encoder import decorated_channels
            if decorated_channels(self.F, deco) != v.vfe_layers[0].linear.in_features:
                raise RuntimeError(f'points have {self.F} features (+ decorations = '
                                   f'{decorated_channels(self.F, deco)}) but the voxel encoder '
                                   f'expects {v.vfe_layers[0].linear.in_features} input channels')
            for i, (lay, (w, sc, sh)) in enumerate(zip(v.vfe_layers, self.vfe_params)):
                chk(L.sf_vfe_layer(
                    cur.data_ptr(), cur_c, self.mp, self.num_points.data_ptr(),
                    s0['indices'].data_ptr(), c0, s0['bound'], deco if i == 0 else 0,
                    _lib.f3(v.voxel_size), _lib.f6(list(v.point_cloud_range)), w.data_ptr(),
                    lay.linear.out_features, sc.data_ptr(), sh.data_ptr(),
                    2 if lay.cat_max else 0, self.vfe_bufs[i].data_ptr(), st))
                cur, cur_c = self.vfe_bufs[i], 2 * lay.linear.out_features
            x = self.vfe_bufs[-1]
        if voxel_features is not None:
            _lib.require(voxel_features, torch.float32, 'voxel_features')
            if voxel_features.shape[0] < s0['bound'] or voxel_features.shape[1] != self.cin_pad:
                raise RuntimeError('voxel_features must be (>= B*max_voxels, padded in_channels)')
            _lib.ptr(voxel_features)
            x = voxel_features
        self._fstate = (x, x_ch, None)

    def _features(self, lo, hi, rb_ready):
        """Layers lo..hi-1 on the current stream; waits for the event of a rulebook in `rb_ready`
        before the first layer that uses it."""
        L, dev = self.L, self.dev
        st = _lib.stream(dev)
        chk = _lib.check
        main = torch.cuda.current_stream(dev)
        c0 = self.counts.data_ptr()
        x, x_ch, block_in = self._fstate
        waited = set()
        for li in range(lo, hi):
            lay = self.layers[li]
            rb = lay.rb
            bufs = self.fbufs[(rb['dst'], lay.cout, lay.out_dtype)]
            busy = {x.data_ptr()}
            if block_in is not None:
                busy.add(block_in.data_ptr())
            out = next(t for t in bufs if t.data_ptr() not in busy)  # 3 buffers: always one free
            residual = None
            if lay.residual_from == 'save':
                block_in = x
            elif lay.residual_from == 'add':
                residual = block_in
            cp = (L.sf_conv_padded_cin_bf16 if lay.bf16 else L.sf_conv_padded_cin)(lay.cin)
            assert cp == x_ch, (cp, x_ch)
            ri = self.rulebooks.index(rb)
            if ri in rb_ready and ri not in waited:
                main.wait_event(rb_ready[ri])
                waited.add(ri)
            k_first = (rb['K'] - 1) // 2 if rb['subm'] else -1
            res_ptr = residual.data_ptr() if residual is not None else None
            if lay.bf16:
                chk(L.sf_sparse_conv_bf16(
                    x.data_ptr(), x.shape[0], cp, rb['nbr'].data_ptr(), rb['stride'], rb['K'],
                    k_first, lay.packed.data_ptr(), lay.cout, lay.scale.data_ptr(),
                    lay.shift.data_ptr(), res_ptr, 1 if lay.relu else 0, out.data_ptr(),
                    1 if lay.out_dtype == torch.float32 else 0, c0 + 4 * rb['dst'],
                    self.sets[rb['dst']]['bound'], st))
            elif lay.ordered:
                chk(L.sf_sparse_conv_f32_ordered(
                    x.data_ptr(), x.shape[0], cp, rb['nbr'].data_ptr(), rb['stride'], rb['K'],
                    k_first, rb['order'].data_ptr(), lay.packed.data_ptr(), lay.cout,
                    lay.scale.data_ptr(), lay.shift.data_ptr(), res_ptr, 1 if lay.relu else 0,
                    out.data_ptr(), c0 + 4 * rb['dst'], self.sets[rb['dst']]['bound'], st))
            else:
                chk(L.sf_sparse_conv_f32(
                    x.data_ptr(), x.shape[0], cp, rb['nbr'].data_ptr(), rb['stride'], rb['K'], k_first,
                    lay.packed.data_ptr(), lay.cout, lay.scale.data_ptr(), lay.shift.data_ptr(),
                    res_ptr, 1 if lay.relu else 0, out.data_ptr(), c0 + 4 * rb['dst'],
                    self.sets[rb['dst']]['bound'], st))
            if residual is not None:
                block_in = None
            x, x_ch = out, lay.cout
            if li == 0 and self.cast0 is not None:
                chk(L.sf_cast_f32_bf16(out.data_ptr(), lay.cout, self.cast0.data_ptr(),
                                       self.cast0.shape[1], c0 + 4 * rb['dst'],
                                       self.sets[rb['dst']]['bound'], st))
                x, x_ch = self.cast0, self.cast0.shape[1]
        self._fstate = (x, x_ch, block_in)

    def _dense_and_after(self):
        """Dense BEV map of the last layer's rows, then the optional SECOND / FPN / height maps."""
        L, dev = self.L, self.dev
        st = _lib.stream(dev)
        chk = _lib.check
        c0 = self.counts.data_ptr()
        s0 = self.sets[0]
        x = self._fstate[0]
        o = self.sets[self.out_set]
        D, H, W = o['shape']
        if getattr(self, '_dense_marked', False):     # the front of this capture built the map
            chk(L.sf_dense_bev_write(x.data_ptr(), c0 + 4 * self.out_set, o['bound'],
                                     self.out_channels, self.B, D, H, W, self.bev.data_ptr(), st))
        else:
            chk(L.sf_dense_bev(x.data_ptr(), o['indices'].data_ptr(), c0 + 4 * self.out_set, o['bound'],
                               self.out_channels, self.B, D, H, W, self.bev.data_ptr(), st))
        if self.bev_backbone is None:
            return self.bev
        feats = self.bev_backbone(self.bev)
        if self.neck is None:
            self.bev_out = feats
        elif self.osf is None:
            self.bev_out = self.neck(feats)[0]
        else:
            # x[0] = cat([neck(x)[0], min_voxel_height[:, None], max_voxel_height[:, None]], 1)
            # (sparsefusion.py:127-130) without the concatenation copy of the big map
            gx, gy = s0['shape'][2], s0['shape'][1]
            nx, ny = gx // self.osf, gy // self.osf
            if self.min_height is None:
                self.min_height = torch.empty((self.B, ny, nx), dtype=torch.float32, device=dev)
                self.max_height = torch.empty_like(self.min_height)
            chk(L.sf_voxel_heights(self.voxels.data_ptr(), s0['indices'].data_ptr(), c0,
                                   s0['bound'], self.mp, self.F, self.B, ny, nx, self.osf,
                                   self.min_height.data_ptr(), self.max_height.data_ptr(), st))
            self.bev_out = self.neck(feats, extra_channels=2)[0]
            c = self.bev_out.shape[1] - 2
            if tuple(self.bev_out.shape[2:]) != (ny, nx):
                raise RuntimeError(f'neck map {tuple(self.bev_out.shape[2:])} vs height map {(ny, nx)}')
            self.bev_out[:, c].copy_(self.min_height)
            self.bev_out[:, c + 1].copy_(self.max_height)
        return self.bev_out

    def _row_order(self, rb, n_dev, n_bound, stream):
        """Mask order of the rulebook's output rows, right behind its neighbour table (same stream)."""
        if 'order' in rb:
            _lib.check(self.L.sf_row_order_by_mask(
                rb['nbr'].data_ptr(), rb['stride'], rb['K'], (rb['K'] - 1) // 2 if rb['subm'] else -1, n_dev, n_bound,
                rb['order'].data_ptr(),
                rb['order_ws'].data_ptr(), rb['order_ws_bytes'], stream))

    def __call__(self, points_list, voxel_features=None):
        return self.enqueue(points_list, voxel_features)

    # ------------------------------------------------------------------
    def results(self):
        """Host view of the last batch (one sync): counts per coordinate set, overflow flag and
        the Voxelization outputs sliced to voxel_num."""
        c = self.counts.tolist()
        n0 = c[0]
        out = dict(counts=c[:-1], overflow=c[-1], coors=self.sets[0]['indices'][:n0],
                   num_points=self.num_points[:n0], voxel_features=self.mean[:n0, :self.in_channels],
                   out_indices=self.sets[self.out_set]['indices'][:c[self.out_set]])
        if self.voxels is not None:
            out['voxels'] = self.voxels[:n0]
        return out

    def capture(self, points_list, voxel_features=None):
        """Capture one batch into a HIP graph bound to these input tensors; returns the graph
        (call .replay()).  The first (eager) call warms up lazy state."""
        self.enqueue(points_list, voxel_features)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.enqueue(points_list, voxel_features)
        return g

    def capture_split(self, points_list, streams, cut=None, voxel_features=None):
        """The same batch as a CHAIN of linear HIP graphs, each on a stream the caller names,
        instead of one graph with forks inside.  Segments: the front (voxelizer, hash, every
        rulebook and mask order), then the layer ranges [0, cut), [cut, end) (+ the dense map and
        what follows it); `cut` may also be a list of ascending layer indices (more ranges).
        streams[i] runs segment i; consecutive segments on the same stream become one graph.  cut
        defaults to the first layer with >= 64 input channels: in front of it the kernels are bound
        by load latency, behind it by the matrix pipe.  Returns an object with .replay().

        Why (tools/queue_probe.py): the same captured frames measured 2.02 / 2.17 /
        2.24 / 2.46 ms per frame with two frames in flight, depending only on which hardware queue
        (GPU_MAX_HW_QUEUES = 4) the HIP runtime had given the branches of each graph.  With plain
        streams the caller decides.  The layout that reproduces the best case by construction: the
        wide layers of ALL frames in flight on ONE stream (one behind the other, each with the
        whole GPU), front + narrow layers on one of TWO other streams by frame parity, four
        LidarBackbone objects in flight (`streams.concurrent_streams` finds streams that do not
        share a queue): the latency-bound kernels of the next two frames fill the matrix-bound
        kernels' shadow and are never in front of them in a queue."""
        dev = self.dev
        nl = len(self.layers)
        if cut is None:
            # (a 64-channel INPUT layer, as behind the Waymo config's HardVFE, is not the cut)
            first_narrow = next((i for i, lay in enumerate(self.layers) if lay.cin < 64), nl)
            cut = next((i for i, lay in enumerate(self.layers) if i > first_narrow and lay.cin >= 64),
                       max(nl // 2, 1))
        cuts = [cut] if isinstance(cut, int) else list(cut)
        # a last cut == number of layers gives the dense map (and what follows it) a segment of its own
        tail = bool(cuts) and cuts[-1] == nl
        edges = [0] + (cuts[:-1] if tail else cuts) + [nl]
        if any(b <= a for a, b in zip(edges, edges[1:])):
            raise ValueError(f'cuts {cuts} must ascend strictly inside the {nl} layers')
        if tail:
            edges.append(nl)
        if len(streams) != len(edges):
            raise ValueError(f'{len(edges)} segments (front + {len(edges) - 1} layer ranges) need as many streams')
        torch.cuda.synchronize(dev)      # earlier batches of this object may still be in flight
        self.enqueue(points_list, voxel_features)
        torch.cuda.synchronize(dev)
        self._dense_marked = False

        def front():
            if self.fused_front:
                self._front_fused(points_list, voxel_features)
                # the cell -> row map of the dense writer needs the output coordinates only: built
                # here, beside the wide layers of the frame before, not behind those of this one
                o = self.sets[self.out_set]
                D, H, W = o['shape']
                _lib.check(self.L.sf_dense_bev_mark(
                    o['indices'].data_ptr(), self.counts.data_ptr() + 4 * self.out_set, o['bound'],
                    self.out_channels, self.B, D, H, W, self.bev.data_ptr(), _lib.stream(dev)))
                self._dense_marked = True
                return
            self._front(points_list, voxel_features)
            ss = _lib.stream(dev)
            for ri in range(1, len(self.rulebooks)):
                self._rulebook(ri, ss)

        def layers(lo, hi):
            def fn():
                if lo == 0:
                    self._feature_input(voxel_features)
                if hi > lo:
                    self._features(lo, hi, {})
                if hi == nl and (lo == nl or not tail):
                    self._dense_and_after()
            return fn

        fns = [front] + [layers(a, b) for a, b in zip(edges, edges[1:])]
        graphs, gstreams, i = [], [], 0
        while i < len(fns):
            j = i + 1
            while j < len(fns) and streams[j] is streams[i]:
                j += 1
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for fn in fns[i:j]:
                    fn()
            graphs.append(g)
            gstreams.append(streams[i])
            i = j
        return _SplitFrame(self, graphs, gstreams)


class FramePipeline:
    """Arbitrary frames through CAPTURED graphs, several in flight (a serving / evaluation loop).

    `LidarBackbone.capture*` binds a graph to its input tensors and to their point count.  This
    class gives every LidarBackbone object in flight a static point buffer of `max_points` rows: a
    frame is copied into it, the rows behind it hold a point just outside `point_cloud_range`
    (the voxelizer rejects it exactly as it rejects any out-of-range point of a real sweep,
    voxelization_cpu.cpp:32-37,71), and the object's split graphs -- captured once, on its first
    frame -- are replayed.  Layout as in bench.py: front + narrow layers on one of two streams by
    frame parity, the wide layers of every frame on one stream, `in_flight` objects.

        pipe = FramePipeline(cfg['pts_voxel_layer'], encoder, 'cuda:0', max_points=400_000, num_features=5)
        t = pipe.submit(points)            # (n, 5) f32 tensor, host or device; returns a ticket
        bev = pipe.result(t)               # waits for that frame only; valid until `in_flight` more
                                           # frames have been submitted (the object is reused)
    """

    def __init__(self, voxel_layer_cfg, encoder, device, max_points, num_features, in_flight=4,
                 **backbone_kw):
        from .streams import concurrent_streams
        self.dev = torch.device(device)
        self.max_points, self.F = int(max_points), int(num_features)
        if in_flight < 1:
            raise ValueError('in_flight >= 1')
        if backbone_kw.get('batch_size', 1) != 1:
            raise NotImplementedError('FramePipeline runs one sample per frame')
        (self.light0, self.light1, self.wide), self.hardware_queues = concurrent_streams(self.dev, 3)
        rng = voxel_layer_cfg['point_cloud_range']
        vs = voxel_layer_cfg['voxel_size']
        # x beyond the upper bound by ten voxels: floor((x - lo) / vs) >= grid, rejected; finite, exact
        sentinel = torch.zeros(self.F, dtype=torch.float32)
        sentinel[0] = float(rng[3]) + 10.0 * float(vs[0])
        self._sentinel = sentinel.to(self.dev)
        self.lanes = []
        for _ in range(in_flight):
            bb = LidarBackbone(voxel_layer_cfg, encoder, self.dev, self.max_points, batch_size=1,
                               **backbone_kw)
            buf = self._sentinel.repeat(self.max_points, 1).contiguous()
            self.lanes.append(dict(bb=bb, buf=buf, valid=0, frame=None, ticket=-1))
        self._next = 0

    def submit(self, points):
        """Enqueue one frame; returns its ticket.  No host synchronisation when `points` is a
        device tensor (a pageable host tensor makes the copy itself synchronous)."""
        if points.dim() != 2 or points.shape[1] != self.F or points.dtype != torch.float32:
            raise RuntimeError(f'points must be (n, {self.F}) float32')
        n = int(points.shape[0])
        if n > self.max_points:
            raise RuntimeError(f'{n} points > max_points {self.max_points}')
        ticket = self._next
        self._next += 1
        lane = self.lanes[ticket % len(self.lanes)]
        bb, buf = lane['bb'], lane['buf']
        light = (self.light0, self.light1)[ticket % 2]
        # the object's previous frame still reads the buffer until its front is done; waiting for
        # the whole frame is the same event the replay waits for anyway
        if bb._last_segment_done is not None:
            light.wait_event(bb._last_segment_done)
        if points.is_cuda:
            # `points` may still be being written by work the caller queued on ITS stream, and the
            # caching allocator may hand its memory out again once the caller drops it: order the
            # copy behind the producer and tell the allocator that `light` reads the tensor
            light.wait_stream(torch.cuda.current_stream(self.dev))
            points.record_stream(light)
        with torch.cuda.stream(light), torch.no_grad():
            if n:
                buf[:n].copy_(points, non_blocking=True)
            if lane['valid'] > n:
                buf[n:lane['valid']] = self._sentinel
            lane['valid'] = n
        if lane['frame'] is None:      # first frame of this object: capture (host-synchronous)
            with torch.no_grad():
                lane['frame'] = bb.capture_split([buf], [light, light, self.wide])
        lane['frame'].streams[0] = light     # (an odd in_flight alternates an object's front stream)
        with torch.cuda.stream(self.wide):
            lane['frame'].replay()
        lane['ticket'] = ticket
        return ticket

    def result(self, ticket):
        """BEV tensor of frame `ticket` (waits for it).  The tensor is the object's static output:
        read or copy it before `in_flight` further frames have been submitted."""
        lane = self.lanes[ticket % len(self.lanes)]
        if lane['ticket'] != ticket:
            raise RuntimeError(f'frame {ticket} is gone: its object now holds frame {lane["ticket"]}')
        lane['bb']._last_segment_done.synchronize()
        bb = lane['bb']
        return bb.bev_out if bb.bev_backbone is not None else bb.bev

    def results(self, ticket):
        """`LidarBackbone.results()` of frame `ticket` (counts, coors, voxels ... of the frame)."""
        self.result(ticket)
        return self.lanes[ticket % len(self.lanes)]['bb'].results()

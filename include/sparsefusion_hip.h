/*
 * sparsefusion_hip.h -- C ABI of libsparsefusion_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for SparseFusion's LiDAR hot path: point-cloud voxelization and the
 * SparseEncoder sparse-3D-convolution backbone.  Plain pointers and sizes only; every pointer
 * named *_dev / points / voxels / ... is DEVICE memory owned by the caller (the library never
 * allocates or frees device memory); `stream` is a hipStream_t passed as void*.  Functions
 * return SF_OK (0) or a negative SF_ERR_* code; sf_last_error() describes the last failure of
 * the calling thread.  Unless stated otherwise a call only enqueues work on `stream`.
 *
 * Row counts that are data dependent live in device memory: an argument pair
 * (`const int *n_dev`, `int n_bound`) means "process *n_dev rows (read on the device when the
 * kernel runs); n_bound is a host-known upper bound used for sizing grids and checking
 * buffers".  n_dev may be NULL, then exactly n_bound rows are processed.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the upstream
 * SparseFusion checkout).
 */
#ifndef SPARSEFUSION_HIP_H_
#define SPARSEFUSION_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_OK 0
#define SF_ERR_BAD_ARG (-1)   /* null pointer, bad size, unsupported parameter               */
#define SF_ERR_WORKSPACE (-2) /* workspace / output buffer smaller than required             */
#define SF_ERR_CAPACITY (-3)  /* coordinate key space does not fit 32 bits, or table overflow */
#define SF_ERR_HIP (-4)       /* a HIP runtime call failed                                   */

#define SF_API __attribute__((visibility("default")))

/* Library version (major*10000 + minor*100 + patch). */
SF_API int sf_version(void);
/* Message for the last error returned on this thread ("" if none). */
SF_API const char *sf_last_error(void);

/* ------------------------------------------------------------------ voxelization ---- */

/* Replaces voxel_layer.dynamic_voxelize (mmdet3d/ops/voxel/src/voxelization.cpp:8,
 * voxelization.h:71-83; kernel voxelization_cuda.cu:24-61, CPU voxelization_cpu.cpp:7-41).
 * points (n,f) f32 with xyz first -> coors (n,3) i32 (z,y,x); all three -1 for a point outside
 * the range (as the CPU path does).  grid = round((hi-lo)/vs) in fp32. */
SF_API int sf_dynamic_voxelize(const float *points, int n, int f, const float voxel_size[3],
                               const float coors_range[6], int *coors, void *stream);

/* Bytes of scratch sf_hard_voxelize needs for n points. */
SF_API size_t sf_hard_voxelize_workspace(int n, int max_points);

/* Replaces voxel_layer.hard_voxelize (voxelization.cpp:7, voxelization.h:51-69; GPU
 * voxelization_cuda.cu:184-326, CPU voxelization_cpu.cpp:43-142) and, optionally, the
 * HardSimpleVFE mean (mmdet3d/models/voxel_encoders/voxel_encoder.py:42-44) and the per-sample
 * glue of detectors/sparsefusion.py:146-157.
 *
 * Semantics are the reference's sequential ones, bit for bit: voxels are numbered in order of
 * their first point; a point that would open voxel number max_voxels stops the scan (it and
 * every later point are dropped); each voxel keeps its first max_points points in point order.
 *
 *  voxels   (rows, max_points, f) f32 : rows [base, base+voxel_num) are written completely
 *                                        (unused slots zero).  May be NULL (skip).
 *  coors    (rows, coors_stride) i32   : coors_stride 3 -> (z,y,x); 4 -> (batch_idx,z,y,x).
 *  num_points_per_voxel (rows) i32     : may be NULL.
 *  mean     (rows, mean_stride) f32    : per-voxel mean of the first mean_features point
 *                                        features, columns [mean_features, mean_stride) zero.
 *                                        May be NULL.
 *  row_count_dev : device int, in = base row of this sample, out = base + voxel_num, so that
 *                  consecutive samples of a batch append without a host round trip.
 *  rows_capacity : rows the four outputs hold; base + voxel_num beyond it is clipped and
 *                  reported through *overflow_dev (if non-NULL, set to 1).
 *  voxel_num_host: if non-NULL the stream is synchronised and this sample's voxel_num is
 *                  returned (the reference's return value); NULL keeps the call asynchronous.
 */
SF_API int sf_hard_voxelize(const float *points, int n, int f, const float voxel_size[3],
                            const float coors_range[6], int max_points, int max_voxels,
                            float *voxels, int *coors, int coors_stride, int batch_idx,
                            int *num_points_per_voxel, float *mean, int mean_features,
                            int mean_stride, int *row_count_dev, int rows_capacity,
                            int *overflow_dev, int *voxel_num_host, void *workspace,
                            size_t workspace_bytes, void *stream);

/* Bytes of the static device scratch behind sf_hard_voxelize_static (per GPU). */
#define SF_STATIC_SCRATCH_BYTES (36u << 20)

/* sf_hard_voxelize for callers that CANNOT ALLOCATE device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference (its op
 * allocates its temporaries through ATen on every call, voxelization_cuda.cu:196-216).  Same
 * kernels, same results, same arguments minus the workspace and minus voxel_num_host (read the
 * count from *row_count_dev): the scratch is a file-scope device array of SF_STATIC_SCRATCH_BYTES
 * that every GPU the library runs on holds until the process ends.  The call performs no
 * hipMalloc / hipFree and no host synchronisation.
 *  Limit: sf_hard_voxelize_workspace(n, max_points) <= SF_STATIC_SCRATCH_BYTES (about 476 000
 *  points at max_points 10, i.e. a nuScenes-shaped 10-sweep frame), else SF_ERR_WORKSPACE.
 *  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two
 *  host threads) race; serialise them.  Consecutive calls are independent (the scratch is
 *  re-initialised by each call).
 *  The FIRST call on a device resolves the array's device address (hipGetSymbolAddress, under a
 *  mutex; not an allocation) and keeps it: make that call eagerly, outside a capture; every later
 *  call on the device may be captured (tests/test_zz_no_alloc.py replays one). */
SF_API int sf_hard_voxelize_static(const float *points, int n, int f, const float voxel_size[3],
                                   const float coors_range[6], int max_points, int max_voxels,
                                   float *voxels, int *coors, int coors_stride, int batch_idx,
                                   int *num_points_per_voxel, float *mean, int mean_features,
                                   int mean_stride, int *row_count_dev, int rows_capacity,
                                   int *overflow_dev, void *stream);

/* Replaces HardSimpleVFE.forward (mmdet3d/models/voxel_encoders/voxel_encoder.py:27-44) on an
 * already materialised voxel tensor: out[v][k] = sum_s voxels[v][s][k] / num_points[v] for
 * k < num_features, columns [num_features, out_stride) zero. */
SF_API int sf_voxel_mean(const float *voxels, const int *num_points_per_voxel, const int *m_dev,
                         int m_bound, int max_points, int f, int num_features, float *out,
                         int out_stride, void *stream);

/* Replaces one VFELayer of HardVFE (mmdet3d/models/voxel_encoders/voxel_encoder.py:383-450 for the
 * decorations and the padding mask, mmdet3d/models/voxel_encoders/utils.py:61-96 for the layer):
 * Linear(bias=False) -> BatchNorm1d (eval, folded into scale/shift) -> ReLU -> max over the
 * max_points slots, in one kernel.
 *  inputs (m, max_points, cin_raw) f32; weight (cout, cin) f32 = nn.Linear.weight with
 *  cin = cin_raw + 3*[bit0] + (3 or 2)*[bit1] + [bit2].
 *  decorate bits (first layer only): 1 cluster-centre offsets, 2 voxel-centre offsets (needs
 *  coors4 (b,z,y,x), voxel_size, coors_range), 4 distance to origin, 8 zero the slots >= num_points,
 *  16 the voxel-centre offset has x and y only, 32 it also overwrites raw x, y before the distance
 *  is taken -- 16 / 32 make the layer the first PFNLayer of PillarFeatureNet
 *  (mmdet3d/models/voxel_encoders/pillar_encoder.py:91-150 with legacy = bit 32; PFNLayer:
 *  voxel_encoders/utils.py:106-180, last_layer -> mode 0, else mode 2).
 *  mode 0: out (m, cout) = max over slots (max_out, not cat_max); 1: out (m, max_points, cout)
 *  point-wise (max_out False); 2: out (m, max_points, 2*cout) = [point-wise, repeated max]. */
SF_API int sf_vfe_layer(const float *inputs, int cin_raw, int max_points, const int *num_points,
                        const int *coors4, const int *m_dev, int m_bound, int decorate,
                        const float voxel_size[3], const float coors_range[6], const float *weight,
                        int cout, const float *scale, const float *shift, int mode, float *out,
                        void *stream);

/* Replaces SparseFusionDetector.extract_voxel_heights (mmdet3d/models/detectors/sparsefusion.py:73-112):
 * min_height / max_height (batch, ny, nx) f32 = per BEV cell (y / out_size_factor,
 * x / out_size_factor) minimum / maximum point z over the voxels in the cell, a z of exactly 0
 * counting as padding; cells without points keep the reference's sentinels +100 / -200.
 * coors4 rows are (b, z, y, x); the sentinel fill is part of the call. */
SF_API int sf_voxel_heights(const float *voxels, const int *coors4, const int *m_dev, int m_bound,
                            int max_points, int f, int batch, int ny, int nx, int out_size_factor,
                            float *min_height, float *max_height, void *stream);

/* ------------------------------------------------------------------ rulebook --------- */

/* Slots (power of two) a coordinate hash table needs for up to n_bound distinct coordinates.
 * A table is `capacity` uint32 keys plus `capacity` int32 values, both caller allocated. */
SF_API int sf_hash_capacity(long long n_bound);

/* Coordinate hash of the active set: indices (n,4) i32 (b,z,y,x) -> table mapping the linear
 * index ((b*D+z)*H+y)*W+x to the row number (first occurrence wins for duplicate rows).
 * Plays the role of the dense `gridOut` of spconv 1.x (mmdet3d/ops/spconv/include/spconv/
 * spconv_ops.h:60-62, indice.cu.h:147-164 prepareSubMGridKernel) without B*D*H*W memory.
 * keys/vals are (re)initialised here. */
SF_API int sf_hash_build(const int *indices, const int *n_dev, int n_bound, const int shape[3],
                         int batch, uint32_t *keys, int *vals, int capacity, int *overflow_dev,
                         void *stream);

/* Bytes of scratch sf_conv_output_set needs. */
SF_API size_t sf_conv_output_set_workspace(int n_bound, int out_capacity);

/* Output active set of a strided SparseConv3d: replaces the output half of
 * getIndicePairsConv (include/spconv/geometry.h:144-194; GPU P1 + unique + P2,
 * indice.cu.h:22-65,112-145).  Output rows come out in the CPU reference's first-touch order
 * (inputs in row order, each input's outputs in getValidOutPos order, geometry.h:57-83).
 *  out_keys/out_vals : hash of the OUTPUT set (coordinate -> output row), built here; it is
 *                      the input table of the next layer.
 *  outids (rows,4)   : (b,z,y,x) of every output row; *n_out_dev receives the row count. */
SF_API int sf_conv_output_set(const int *indices, const int *n_dev, int n_bound, int batch,
                              const int out_shape[3], const int ksize[3], const int stride[3],
                              const int padding[3], uint32_t *out_keys, int *out_vals,
                              int out_capacity, int *outids, int outids_rows, int *n_out_dev,
                              int *overflow_dev, void *workspace, size_t workspace_bytes,
                              void *stream);

/* Bytes of scratch sf_strided_rulebook needs. */
SF_API size_t sf_strided_rulebook_workspace(int n_bound, int out_capacity, const int ksize[3],
                                            const int stride[3]);

/* sf_conv_output_set and sf_neighbor_table of a strided SparseConv3d in one call (same
 * reference functions: getIndicePairsConv, geometry.h:144-194), built from the INPUT side: every
 * input remembers the output-table slots it touched, so the table is probed once per pair instead
 * of once per (output row, kernel offset) -- most of which miss at stride 2.  Results are
 * identical to the two separate calls (same output row order, same table; for duplicate input
 * coordinates the first occurrence wins).  nbr is (K, nbr_stride) with nbr_stride >= outids_rows. */
SF_API int sf_strided_rulebook(const int *indices, const int *n_dev, int n_bound, int batch,
                               const int out_shape[3], const int ksize[3], const int stride[3],
                               const int padding[3], uint32_t *out_keys, int *out_vals,
                               int out_capacity, int *outids, int outids_rows, int *n_out_dev,
                               int *overflow_dev, int *nbr, int nbr_stride, void *workspace,
                               size_t workspace_bytes, void *stream);

/* Neighbour table = the rulebook in output-major form: nbr[k*nbr_stride + o] is the input row
 * that feeds output row o through kernel offset k = (kz*kH+ky)*kW+kx, or -1.  Same pairs as
 * indicePairs/indiceNum of getIndicePairsSubM / getIndicePairsConv (geometry.h:247-297,
 * :144-194): in = out*stride - padding + (kz,ky,kx).  For SubMConv3d pass outids = the input
 * indices, stride 1, padding k/2 and identity_center=1 (centre offset pairs every row with
 * itself, spconv_ops.h:300-303). */
SF_API int sf_neighbor_table(const int *outids, const int *n_out_dev, int n_out_bound, int batch,
                             const int in_shape[3], const int ksize[3], const int stride[3],
                             const int padding[3], const uint32_t *in_keys, const int *in_vals,
                             int in_capacity, int identity_center, int *nbr, int nbr_stride,
                             void *stream);

/* Bytes of scratch sf_export_indice_pairs needs. */
SF_API size_t sf_export_indice_pairs_workspace(int K, int n_in);

/* Neighbour table -> the reference's rulebook tensors: indice_pairs (K,2,n_in) i32 padded with
 * -1 and indice_num (K) i32, exactly the layout of sparse_conv_ext.get_indice_pairs_3d
 * (mmdet3d/ops/spconv/src/all.cc:21-51, spconv_ops.h:55-59), pairs of each offset listed in
 * ascending input-row order (the CPU reference's order, geometry.h:171-192). */
SF_API int sf_export_indice_pairs(const int *nbr, int nbr_stride, int K, const int *n_out_dev,
                                  int n_out_bound, int n_in, int *indice_pairs, int *indice_num,
                                  void *workspace, size_t workspace_bytes, void *stream);

/* The inverse: the reference's rulebook tensors (indice_pairs (K,2,L), indice_num (K), e.g. from
 * the CPU path of get_indice_pairs) -> neighbour table nbr (K, nbr_stride), so that
 * sf_sparse_conv_f32 can run from a rulebook in the reference's own format
 * (the argument list of sparse_conv_ext.indice_conv_fp32, all.cc / ops.py:108-126). */
SF_API int sf_indice_pairs_to_table(const int *indice_pairs, const int *indice_num, int K, int L,
                                    int n_out, int *nbr, int nbr_stride, void *stream);

/* ---- the coordinate phase of a frame in one call (csrc/front.hip) -------------------- */

/* One coordinate set of the encoder: the voxels (set 0) or the output rows of a strided layer. */
typedef struct SfCoordSet {
  int shape[3];      /* spatial shape (D, H, W) */
  int bound;         /* rows `indices` holds (host bound on the row count) */
  int *indices;      /* (bound, 4) i32 (b,z,y,x) */
  uint32_t *keys;    /* coordinate hash: `capacity` keys ... */
  int *vals;         /* ... and values (sf_hash_capacity(bound) slots) */
  int capacity;
} SfCoordSet;

/* One rulebook: a SubM layer geometry on a set (dst == src) or a strided one between two sets. */
typedef struct SfRulebook {
  int src, dst;      /* indices into SfFrontDesc.sets */
  int ksize[3], stride[3], padding[3];
  int subm;
  int *nbr;          /* (K, nbr_stride) neighbour table, nbr_stride >= sets[dst].bound */
  int nbr_stride;
  int *plan;         /* sf_row_order_by_mask plan of the table (sf_row_order_size ints) or NULL */
  void *plan_ws;     /* its workspace (sf_row_order_workspace bytes) */
  size_t plan_ws_bytes;
  int *first;        /* strided only: sets[dst].capacity ints of scratch of this rulebook */
} SfRulebook;

typedef struct SfFrontDesc {
  /* voxelizer (arguments of sf_hard_voxelize, one point cloud per sample) */
  int batch;
  const float *const *points;   /* host array of `batch` device pointers, (n_b, num_features) f32 */
  const int *num_points;        /* host array of `batch` point counts */
  int num_features;
  float voxel_size[3], coors_range[6];
  int max_points, max_voxels;
  float *voxels;                /* (sets[0].bound, max_points, num_features) or NULL */
  int *num_points_per_voxel;    /* (sets[0].bound) or NULL */
  float *mean;                  /* (sets[0].bound, mean_stride) or NULL */
  int mean_features, mean_stride;
  void *vox_workspace;          /* batch * vox_workspace_bytes bytes, a slice per sample ... */
  size_t vox_workspace_bytes;   /* ... of sf_hard_voxelize_workspace(max n_b, max_points) bytes */
  /* coordinate sets and rulebooks of the encoder */
  int n_sets;
  const SfCoordSet *sets;
  int *counts_dev;              /* n_sets + 1 device ints: row count per set, then the overflow flag */
  int n_rulebooks;
  const SfRulebook *rulebooks;  /* in layer order */
  void *rb_workspace;           /* max over the strided rulebooks of sf_strided_rulebook_workspace */
  size_t rb_workspace_bytes;
} SfFrontDesc;

/* Voxelizer + coordinate hashes + every rulebook + the plans of the mask-ordered layers of one
 * batch, enqueued by one call: what SparseFusionDetector.voxelize (detectors/sparsefusion.py:
 * 133-161) and the get_indice_pairs of every sparse convolution of SparseEncoder.forward
 * (ops/spconv/conv.py:113-204 -> spconv_ops.h:27-141) compute.  Same kernels and bit-identical
 * buffers as the sequence sf_hard_voxelize (per sample, rows appended, coors_stride 4) ->
 * sf_hash_build -> sf_neighbor_table / sf_strided_rulebook (-> sf_row_order_by_mask) per rulebook,
 * with fewer launches: one fill kernel for all tables, row assignments that open the table entries
 * of their rows, the table scatter of a strided rulebook in one launch with the SubM table of its
 * output set.  The descriptor is read during the call only (host memory; device pointers inside).
 * counts_dev is zeroed by the call. */
SF_API int sf_lidar_front(const SfFrontDesc *desc, void *stream);

/* ------------------------------------------------------------------ convolution ------ */

/* Channel count a feature matrix must be padded to as conv INPUT (8, 16, 32, 64 or 128), or
 * -1 if unsupported.  Output channels must be 16, 32, 64 or 128. */
SF_API int sf_conv_padded_cin(int cin);

/* Re-order filters (K, cin, cout) f32 [= spconv-1.x (kD,kH,kW,Cin,Cout) flattened,
 * mmdet3d/ops/spconv/conv.py:98-99] into the MFMA fragment order the conv kernel streams.
 * w_packed holds K * max(sf_conv_padded_cin(cin),16) * cout floats. */
SF_API int sf_pack_filters_f32(const float *w_kio, int K, int cin, int cout, float *w_packed,
                               void *stream);

/* Replaces sparse_conv_ext.indice_conv_fp32 / indice_subm_conv (all.cc:21-51;
 * indiceConv<float>, spconv_ops.h:260-361: gather -> mm -> scatter-add per offset) with one
 * output-stationary kernel, fused with the layer's epilogue
 *     y = conv(x) * scale + shift (+ residual) ; ReLU
 * i.e. eval-mode BatchNorm1d folded (make_sparse_convmodule, mmdet3d/ops/sparse_block.py:
 * 142-204) and the SparseBasicBlock tail (sparse_block.py:127-137).  scale/shift/residual may
 * be NULL (plain convolution).  features_in is (n_in_bound, cin_padded) f32 -- n_in_bound rows
 * must be addressable: the narrow-channel kernel bounds its gathers by it (tables of 2 GiB and
 * more run on the general kernels) --; out (n_out, cout).
 * k_first >= 0 accumulates that offset first (the reference adds the SubM centre first). */
SF_API int sf_sparse_conv_f32(const float *features_in, int n_in_bound, int cin_padded, const int *nbr,
                              int nbr_stride, int K, int k_first, const float *w_packed, int cout,
                              const float *scale, const float *shift, const float *residual,
                              int relu, float *out, const int *n_out_dev, int n_out_bound,
                              void *stream);

/* ---- bf16 storage / bf16 MFMA / fp32 accumulation (BASELINE.json configs[2]) ----------
 * The reference reaches reduced precision through mmcv's auto_fp16 on SparseEncoder.forward
 * (mmdet3d/models/middle_encoders/sparse_encoder.py:53,95: fp16_enabled casts voxel_features to
 * half and every spconv layer then runs in half); here the reduced type is bf16 and the cast is
 * explicit.  Feature matrices are (n, c) bf16 row-major; scale/shift stay fp32. */

/* Channel count a bf16 feature matrix must be padded to as conv INPUT (16, 32, 64 or 128), or
 * -1 if unsupported. */
SF_API int sf_conv_padded_cin_bf16(int cin);

/* Filters (K, cin, cout) f32 -> bf16 (round to nearest even) in the MFMA fragment order of
 * sf_sparse_conv_bf16.  w_packed holds K * sf_conv_padded_cin_bf16(cin) * cout bf16 values. */
SF_API int sf_pack_filters_bf16(const float *w_kio, int K, int cin, int cout, void *w_packed,
                                void *stream);

/* src (n, c_src) f32 -> dst (n, c_dst) bf16, round to nearest even, columns >= c_src zero: the
 * explicit form of the `.half()` cast auto_fp16 applies to voxel_features (sparse_encoder.py:95). */
SF_API int sf_cast_f32_bf16(const float *src, int c_src, void *dst, int c_dst, const int *n_dev,
                            int n_bound, void *stream);

/* sf_sparse_conv_f32 with bf16 features / filters / residual and fp32 accumulation and epilogue
 * (indiceConv<at::Half>, spconv_ops.h:260-361 + the fused BN/residual/ReLU tail).  features_in is
 * (n_in_bound, cin_padded) bf16 -- n_in_bound rows must be addressable (the gather is bounds
 * checked against it and the table must stay below 2 GiB); out is (n_out, cout) bf16, or fp32
 * if out_is_f32 (the last layer feeds sf_dense_bev).  residual is (n_out, cout) bf16 or NULL. */
SF_API int sf_sparse_conv_bf16(const void *features_in, int n_in_bound, int cin_padded,
                               const int *nbr, int nbr_stride, int K, int k_first,
                               const void *w_packed, int cout, const float *scale,
                               const float *shift, const void *residual, int relu, void *out,
                               int out_is_f32, const int *n_out_dev, int n_out_bound, void *stream);

/* Name of the convolution kernel the calling thread launched last, template arguments included,
 * exactly as a rocprofv3 kernel trace prints it (minus the anonymous namespace and the argument
 * list): bench.py's roofline.kernel.  No reference counterpart (diagnostic). */
SF_API const char *sf_last_conv_kernel(void);

/* Diagnostic hooks exported by the library but NOT part of this interface (no declaration here
 * on purpose): sf_debug_conv_profile, sf_debug_ordered_profile (a device buffer every convolution
 * launch adds in-kernel cycle stamps to), sf_debug_conv_row_order and sf_debug_ordered_grid_cap (a
 * test hook that limits the persistent grid of the ordered kernels).  They set process-wide
 * pointers read by every later launch on every stream: not thread- or stream-safe, used by
 * tools/prof_conv.py / tools/prof_ordered.py / tools/ab_ordered.py only; NULL (the initial state)
 * turns them off. */

/* ---- the same convolution over mask-ordered rows (csrc/spconv_f32_ordered.hip) ---- */

/* Workspace bytes of sf_row_order_by_mask for up to n_bound output rows. */
SF_API size_t sf_row_order_workspace(int n_bound);

/* Number of ints the row_order buffer of sf_row_order_by_mask must hold for n_bound rows. */
SF_API int sf_row_order_size(int n_bound);

/* Builds the PLAN of a rulebook for sf_sparse_conv_f32_ordered in row_order (T = ceil(n_bound/64)):
 *   [0, n_bound)            the output rows sorted (stably) by their K-bit neighbour mask (bit k set
 *                           <=> nbr[k][row] >= 0); entries [*n_dev, n_bound) are -1 or unspecified
 *   [n_bound, +T)           the tile schedule: the 64-row tiles of the sorted rows by descending cost
 *                           ((16-row block, offset) products + 3 per active offset), dealt to the
 *                           persistent workgroups of the convolution
 *   + 32 T                  tile headers: [0] phases np = max(active offsets, 3), [1..np] the phases
 *                           ready made in execution order -- the offset k_first (the centre of a
 *                           SubM rulebook; -1: none) first when active, then ascending, then
 *                           padding: k | block mask << 5 | flags << 9 --, [28] cost, [29] mask of
 *                           the active offsets
 *   + 27 * 64 T             the neighbour table permuted into the sorted order, column per offset
 * The buffer holds sf_row_order_size(n_bound) ints.  The sort is a hand-written 3-pass radix sort
 * bounded by the device-side row count (no library call, no host synchronisation); seven short
 * launches per rulebook.
 * No reference counterpart: the reference's rulebook has no row order of its own (pairs are
 * listed per offset, spconv_ops.h:55-59); this is the schedule sf_sparse_conv_f32_ordered runs
 * its tiles in, built once per rulebook and shared by every layer that uses it. */
SF_API int sf_row_order_by_mask(const int *nbr, int nbr_stride, int K, int k_first, const int *n_dev,
                                int n_bound, int *row_order, void *workspace, size_t workspace_bytes,
                                void *stream);

/* sf_sparse_conv_f32 (same reference interface: indice_conv_fp32 / indice_subm_conv,
 * spconv_ops.h:260-361, + the fused BN / residual / ReLU tail) with a different work
 * decomposition: tiles of 64 entries of row_order (the plan written by sf_row_order_by_mask for
 * the same neighbour table and the same n_out_bound; required), no pair compaction, sums kept in
 * the MFMA accumulators, (offset, 16-row block) products without a partner skipped.  The kernel
 * reads the plan's permuted table; nbr / nbr_stride are only validated.
 * Results are fp32 FMA chains in the reference's order (centre offset first for SubM, then
 * offsets ascending), independent of row_order; the order is part of the plan (k_first of
 * sf_row_order_by_mask), k_first here must be the same value.  features_in is (n_in_bound, cin_padded) f32
 * with cin_padded in {16, 32, 64, 128}: n_in_bound rows must be addressable (the gather is bounds
 * checked against it; table < 4 GiB). */
SF_API int sf_sparse_conv_f32_ordered(const float *features_in, int n_in_bound, int cin_padded,
                                      const int *nbr, int nbr_stride, int K, int k_first,
                                      const int *row_order, const float *w_packed, int cout,
                                      const float *scale, const float *shift, const float *residual,
                                      int relu, float *out, const int *n_out_dev, int n_out_bound,
                                      void *stream);

/* ------------------------------------------------------------------ dense BEV -------- */

/* Replaces SparseConvTensor.dense() + view (mmdet3d/ops/spconv/structure.py:55-64,
 * mmdet3d/models/middle_encoders/sparse_encoder.py:122-125): out (B, C*D, H, W) f32 with
 * channel = c*D + z, zero where inactive (the zero fill is part of this call, done by kernels:
 * like every entry point of this library it captures into a HIP graph as kernel nodes only). */
SF_API int sf_dense_bev(const float *features, const int *indices, const int *n_dev, int n_bound,
                        int channels, int batch, int D, int H, int W, float *out, void *stream);

/* sf_dense_bev in two calls, for a pipeline that knows the output coordinates long before the
 * features (the rulebooks of a frame are built while the previous frame still convolves):
 * sf_dense_bev_mark writes the cell -> row map into the c = 0 planes of `out` (zero fill + one
 * mark per row; needs `indices` only), sf_dense_bev_write then writes every channel plane from
 * `features` (and overwrites the map with the real channel 0 last).  mark, write in this order,
 * with the same (n_dev, n_bound, shape, out), equal sf_dense_bev; `out` must not be touched in
 * between.  Same reference lines as sf_dense_bev. */
SF_API int sf_dense_bev_mark(const int *indices, const int *n_dev, int n_bound, int channels,
                             int batch, int D, int H, int W, float *out, void *stream);
SF_API int sf_dense_bev_write(const float *features, const int *n_dev, int n_bound, int channels,
                              int batch, int D, int H, int W, float *out, void *stream);

/* Zero-pad feature columns: src (n, c_src) -> dst (n, c_dst), c_dst >= c_src. */
SF_API int sf_pad_channels_f32(const float *src, int c_src, float *dst, int c_dst,
                               const int *n_dev, int n_bound, void *stream);

/* ------------------------------------------------------------------ BEV backbone / neck */

/* Bytes of the packed filter image of one dense 2-D layer (0 for a bad shape).  `up` > 1 is a
 * transposed convolution with kernel = stride = up (ksize must then be 1, see below). */
SF_API size_t sf_conv2d_packed_bytes(int cout, int cin, int ksize, int up);

/* Re-orders the filters of one nn.Conv2d (mmdet3d/models/backbones/second.py:38-55,
 * necks/second_fpn.py:54-59; weight (cout, cin, ksize, ksize) f32) into the MFMA fragment order
 * sf_conv2d_f32 streams.  A transposed convolution with kernel = stride = up
 * (second_fpn.py:46-51, nn.ConvTranspose2d weight (cin, cout, up, up)) is handed over as the
 * equivalent up*up 1x1 filters: weight (up*up*cout, cin) with row (dy*cout + co)*up + dx =
 * w[:, co, dy, dx] (dx fastest: a lane's 4 consecutive packed channels are consecutive output
 * x, stored as one 8 / 16-byte piece), ksize = 1.  packed holds sf_conv2d_packed_bytes(cout, cin, ksize, up) bytes. */
SF_API int sf_conv2d_pack_filters_f32(const float *weight, int cout, int cin, int ksize, int up,
                                      float *packed, void *stream);

/* One Conv2d / ConvTranspose2d(k = s) + BatchNorm2d(eval) + ReLU of SECOND / SECONDFPN
 * (second.py:36-60,82-86; second_fpn.py:61-63,86-91) as a single launch, fp32 on the matrix
 * cores (exact fp32 FMA chains), NCHW:
 *   in  (batch, cin, h, w);   out (batch, out_channels, up*ho, up*wo), of which channels
 *   [out_channel_offset, out_channel_offset + cout) are written (the rest untouched: a neck's
 *   deblocks write straight into the concatenated map), ho = (h + 2*pad - ksize)/stride + 1;
 *   out = relu?(conv(in) * scale[co] + shift[co]); scale / shift (cout) or NULL (1 / 0).
 * Supported (ksize, stride): (3,1), (3,2), (1,1), (2,2); dilation 1, groups 1, no bias (fold a
 * bias into shift).  up > 1 needs ksize = 1, stride = 1, pad = 0. */
SF_API int sf_conv2d_f32(const float *in, const float *packed, const float *scale,
                         const float *shift, float *out, int batch, int cin, int h, int w, int cout,
                         int ksize, int stride, int pad, int up, int out_channels,
                         int out_channel_offset, int relu, void *stream);

/* ------------------------------------------------------------------ dynamic scatter -- */

/* Per-dimension maximum of coors (n, ndim) i32 into max_dev[ndim] (device; -1 where n == 0 or all
 * entries negative).  The reference sizes its linear-id space the same way
 * (`coors.max(0) + 1`, mmdet3d/ops/voxel/src/scatter_points_cuda.cu:241). */
SF_API int sf_coors_max(const int *coors, int n, int ndim, int *max_dev, void *stream);

/* Workspace bytes of sf_dynamic_scatter for a coordinate space dims[ndim]; 0 if the space has more
 * than 2^32 cells (or a dimension < 1). */
SF_API size_t sf_dynamic_scatter_workspace(const int *dims, int ndim);

/* Replaces voxel_layer.dynamic_point_to_voxel_forward (mmdet3d/ops/voxel/src/voxelization.h:96-108,
 * scatter_points_cuda.cu:230-311): feats (n, c) f32 and coors (n, ndim) i32 are reduced into the
 * unique non-negative coordinates, output rows in ascending linear-id order (= lexicographic in
 * the coordinates).  reduce_type 0 sum, 1 mean, 2 max (the reference's reduce_t).  Every dims[i]
 * must exceed every coordinate of dimension i.  Outputs, caller allocated for `capacity` rows:
 * voxel_feats (capacity, c), voxel_coors (capacity, ndim), count (capacity) [points per voxel for
 * mean, else zeros, as in the reference], point2voxel (n) [-1 for dropped points];
 * *num_voxels_dev receives the number of voxels, *overflow_dev is raised if it exceeds capacity.
 * Sums use float atomics (order dependent in the last bits, like the reference's; max is exact). */
SF_API int sf_dynamic_scatter(const float *feats, const int *coors, int n, int c, int ndim,
                              const int *dims, int reduce_type, float *voxel_feats,
                              int *voxel_coors, int *point2voxel, int *count, int capacity,
                              int *num_voxels_dev, int *overflow_dev, void *ws, size_t ws_bytes,
                              void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SPARSEFUSION_HIP_H_ */

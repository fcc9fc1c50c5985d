// DynamicScatter forward for gfx950: reduce per-point features into unique voxels.
//
// Reference: dynamic_point_to_voxel_forward_gpu (mmdet3d/ops/voxel/src/scatter_points_cuda.cu:230-311)
// linearises the coordinates (coors_id_kernel, :80-95), argsorts the 64-bit ids, marks segment heads
// (coors_map_init_kernel, :97-115), cumsums them into voxel indices and reduces with float atomics
// (feats_reduce_kernel, :117-150): output voxels come out in ASCENDING linear-id order, points with a
// negative coordinate are dropped (map -1).
//
// Here there is no sort: the linear-id space is a bitmap (one bit per cell, 10.6 MB for the nuScenes
// grid), every valid point sets its bit, a two-level popcount scan turns the bitmap into "number of
// occupied cells before this word", and a point's voxel index is that prefix plus the popcount of
// the lower bits of its own word -- the rank of its id among the occupied ids, i.e. the reference's
// order, deterministically.  HBM-bound: bitmap zero + 2 reads, points read twice.
#include "sf_common.h"

namespace {

constexpr int kScanThreads = 256;
constexpr int kWordsPerThread = 8;
constexpr int kWordsPerBlock = kScanThreads * kWordsPerThread;

__device__ __forceinline__ long long linear_id(const int *__restrict__ c, int ndim,
                                               const int *__restrict__ dims) {
  long long id = 0;
  for (int i = 0; i < ndim; ++i) {
    const int t = c[i];
    if (t < 0) return -1;  // scatter_points_cuda.cu:88-92
    id = id * dims[i] + t;
  }
  return id;
}

__global__ void coors_max_kernel(const int *__restrict__ coors, int n, int ndim,
                                 int *__restrict__ mx) {
  int m[4] = {-1, -1, -1, -1};
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n; x += gridDim.x * blockDim.x)
    for (int i = 0; i < ndim; ++i) {
      int t = coors[(size_t)x * ndim + i];
      m[i] = t > m[i] ? t : m[i];
    }
  for (int i = 0; i < ndim; ++i) {
    int v = m[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      int o = __shfl_xor(v, d, 64);
      v = o > v ? o : v;
    }
    if ((threadIdx.x & 63) == 0 && v >= 0) atomicMax(&mx[i], v);
  }
}

__global__ void ds_setbits_kernel(const int *__restrict__ coors, int n, int ndim, int3 d3,
                                  int d4, unsigned *__restrict__ bitmap) {
  const int dims[4] = {d3.x, d3.y, d3.z, d4};
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n; x += gridDim.x * blockDim.x) {
    long long id = linear_id(coors + (size_t)x * ndim, ndim, dims);
    if (id >= 0) atomicOr(&bitmap[id >> 5], 1u << (id & 31));
  }
}

__global__ void ds_count_kernel(const unsigned *__restrict__ bitmap, long long words,
                                int *__restrict__ bsum) {
  __shared__ int smem[17];
  long long w0 = (long long)blockIdx.x * kWordsPerBlock + threadIdx.x * kWordsPerThread;
  int c = 0;
#pragma unroll
  for (int i = 0; i < kWordsPerThread; ++i)
    if (w0 + i < words) c += __popc(bitmap[w0 + i]);
  int tot = sf_block_sum(c, smem);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ void ds_prefix_kernel(const unsigned *__restrict__ bitmap, long long words,
                                 const int *__restrict__ bsum, int nblocks,
                                 int *__restrict__ wprefix, int *__restrict__ num_voxels) {
  __shared__ int smem[17];
  int before, all;
  sf_block_prefix_of(bsum, nblocks, blockIdx.x, smem, &before, &all);
  long long w0 = (long long)blockIdx.x * kWordsPerBlock + threadIdx.x * kWordsPerThread;
  int pc[kWordsPerThread], c = 0;
#pragma unroll
  for (int i = 0; i < kWordsPerThread; ++i) {
    pc[i] = (w0 + i < words) ? __popc(bitmap[w0 + i]) : 0;
    c += pc[i];
  }
  int tot, ex = sf_block_excl_scan(c, smem, &tot);
  int run = before + ex;
#pragma unroll
  for (int i = 0; i < kWordsPerThread; ++i) {
    if (w0 + i < words) wprefix[w0 + i] = run;
    run += pc[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *num_voxels = all;
}

__global__ void ds_init_kernel(float *__restrict__ vf, int *__restrict__ cnt, long long rows,
                               int c, float init) {
  const long long total = rows * c;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x)
    vf[e] = init;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < rows;
       e += (long long)gridDim.x * blockDim.x)
    cnt[e] = 0;
}

__device__ __forceinline__ void atomic_max_f32(float *addr, float v) {
  // order-preserving integer views of IEEE floats; the slot starts at -inf.  The branch is on the
  // SIGN BIT: -0.0f compares >= 0 but its pattern (INT_MIN) would lose a signed max against -inf
  if (__float_as_int(v) >= 0)
    atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
  else
    atomicMin(reinterpret_cast<unsigned *>(addr), __float_as_uint(v));
}

__global__ void ds_reduce_kernel(const float *__restrict__ feats, const int *__restrict__ coors,
                                 int n, int c, int ndim, int3 d3, int d4,
                                 const unsigned *__restrict__ bitmap,
                                 const int *__restrict__ wprefix, int reduce_type, int capacity,
                                 float *__restrict__ vf, int *__restrict__ vc,
                                 int *__restrict__ p2v, int *__restrict__ cnt,
                                 int *__restrict__ overflow) {
  const int dims[4] = {d3.x, d3.y, d3.z, d4};
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n; x += gridDim.x * blockDim.x) {
    const int *cx = coors + (size_t)x * ndim;
    long long id = linear_id(cx, ndim, dims);
    int to = -1;
    if (id >= 0) {
      unsigned word = bitmap[id >> 5];
      to = wprefix[id >> 5] + __popc(word & ((1u << (id & 31)) - 1u));
    }
    p2v[x] = to;
    if (to < 0) continue;
    if (to >= capacity) {
      *overflow = 1;
      continue;
    }
    for (int i = 0; i < ndim; ++i) vc[(size_t)to * ndim + i] = cx[i];  // same value from every writer
    if (reduce_type == 1) atomicAdd(&cnt[to], 1);
    const float *f = feats + (size_t)x * c;
    float *o = vf + (size_t)to * c;
    if (reduce_type == 2) {
      for (int i = 0; i < c; ++i) atomic_max_f32(&o[i], f[i]);
    } else {
      for (int i = 0; i < c; ++i) atomicAdd(&o[i], f[i]);
    }
  }
}

__global__ void ds_mean_kernel(float *__restrict__ vf, const int *__restrict__ cnt,
                               const int *__restrict__ num_voxels, int capacity, int c) {
  int m = *num_voxels;
  if (m > capacity) m = capacity;
  const long long total = (long long)m * c;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x)
    vf[e] = vf[e] / (float)cnt[e / c];  // reduced_feats /= reduce_count (:303-305)
}

long long cells_of(const int *dims, int ndim) {
  long long cells = 1;
  for (int i = 0; i < ndim; ++i) {
    if (dims[i] < 1) return -1;
    cells *= dims[i];
    if (cells > (1ll << 32)) return -1;
  }
  return cells;
}

}  // namespace

extern "C" int sf_coors_max(const int *coors, int n, int ndim, int *max_dev, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  SF_CHECK_ARG(n >= 0 && ndim >= 1 && ndim <= 4 && max_dev, "bad n / ndim / max_dev");
  SF_FILL_U32(max_dev, ndim, 0xFFFFFFFFu, st);  // -1
  if (n == 0) return SF_OK;
  SF_CHECK_ARG(coors, "coors is NULL");
  hipLaunchKernelGGL(coors_max_kernel, dim3(sf_grid(n, 256)), dim3(256), 0, st, coors, n, ndim,
                     max_dev);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" size_t sf_dynamic_scatter_workspace(const int *dims, int ndim) {
  long long cells = cells_of(dims, ndim);
  if (cells < 0) return 0;
  long long words = (cells + 31) / 32;
  long long blocks = (words + kWordsPerBlock - 1) / kWordsPerBlock;
  return (size_t)(words * 4 + words * 4 + blocks * 4 + 256);
}

extern "C" int sf_dynamic_scatter(const float *feats, const int *coors, int n, int c, int ndim,
                                  const int *dims, int reduce_type, float *voxel_feats,
                                  int *voxel_coors, int *point2voxel, int *count, int capacity,
                                  int *num_voxels_dev, int *overflow_dev, void *ws, size_t ws_bytes,
                                  void *stream) {
  hipStream_t st = (hipStream_t)stream;
  SF_CHECK_ARG(n >= 0 && c >= 1 && ndim >= 1 && ndim <= 4 && capacity >= 0, "bad n / c / ndim");
  SF_CHECK_ARG(reduce_type >= 0 && reduce_type <= 2, "reduce_type must be 0 sum, 1 mean, 2 max");
  SF_CHECK_ARG(dims && num_voxels_dev && overflow_dev, "dims / num_voxels_dev / overflow_dev NULL");
  long long cells = cells_of(dims, ndim);
  if (cells < 0) {
    sf_set_error("coordinate space must have 1..2^32 cells");
    return SF_ERR_CAPACITY;
  }
  size_t need = sf_dynamic_scatter_workspace(dims, ndim);
  if (!ws || ws_bytes < need) {
    sf_set_error("workspace too small: %zu < %zu bytes", ws_bytes, need);
    return SF_ERR_WORKSPACE;
  }
  SF_CHECK_ARG(capacity == 0 || (voxel_feats && voxel_coors && count), "output pointer is NULL");
  SF_CHECK_ARG(n == 0 || (feats && coors && point2voxel), "input pointer is NULL");
  const long long words = (cells + 31) / 32;
  const int blocks = (int)((words + kWordsPerBlock - 1) / kWordsPerBlock);
  unsigned *bitmap = (unsigned *)ws;
  int *wprefix = (int *)(bitmap + words);
  int *bsum = wprefix + words;
  const int3 d3 = make_int3(dims[0], ndim > 1 ? dims[1] : 1, ndim > 2 ? dims[2] : 1);
  const int d4 = ndim > 3 ? dims[3] : 1;
  SF_FILL_U32(bitmap, words, 0u, st);
  if (n > 0) {
    hipLaunchKernelGGL(ds_setbits_kernel, dim3(sf_grid(n, 256)), dim3(256), 0, st, coors, n, ndim,
                       d3, d4, bitmap);
    SF_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(ds_count_kernel, dim3(blocks), dim3(kScanThreads), 0, st, bitmap, words, bsum);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(ds_prefix_kernel, dim3(blocks), dim3(kScanThreads), 0, st, bitmap, words, bsum,
                     blocks, wprefix, num_voxels_dev);
  SF_CHECK_LAUNCH();
  if (capacity > 0) {
    const float init = reduce_type == 2 ? -__builtin_huge_valf() : 0.0f;
    hipLaunchKernelGGL(ds_init_kernel, dim3(sf_grid((long long)capacity * c, 256)), dim3(256), 0, st,
                       voxel_feats, count, (long long)capacity, c, init);
    SF_CHECK_LAUNCH();
  }
  if (n > 0) {
    hipLaunchKernelGGL(ds_reduce_kernel, dim3(sf_grid(n, 256)), dim3(256), 0, st, feats, coors, n, c,
                       ndim, d3, d4, bitmap, wprefix, reduce_type, capacity, voxel_feats,
                       voxel_coors, point2voxel, count, overflow_dev);
    SF_CHECK_LAUNCH();
    if (reduce_type == 1 && capacity > 0) {
      hipLaunchKernelGGL(ds_mean_kernel, dim3(sf_grid((long long)capacity * c, 256)), dim3(256), 0,
                         st, voxel_feats, count, num_voxels_dev, capacity, c);
      SF_CHECK_LAUNCH();
    }
  }
  return SF_OK;
}

// ------------------------------------------------------------------------------------------
// extract_voxel_heights (mmdet3d/models/detectors/sparsefusion.py:73-112): per BEV cell
// (b, y / osf, x / osf) the minimum and maximum point height over the voxels that fall in it;
// a z of exactly 0 is a padding slot (sentinels +100 / -200, :82-86).  The reference loops 40
// height slices with masked scatters; here one thread reduces one voxel's slots and issues two
// order-preserving integer atomics.  HBM-bound: the z column of the voxel tensor is read once.
namespace {

__global__ void heights_init_kernel(float *__restrict__ mn, float *__restrict__ mx, long long cells) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < cells;
       e += (long long)gridDim.x * blockDim.x) {
    mn[e] = 100.0f;
    mx[e] = -200.0f;
  }
}

__device__ __forceinline__ void atomic_min_f32(float *addr, float v) {
  if (v >= 0.0f)
    atomicMin(reinterpret_cast<int *>(addr), __float_as_int(v));
  else
    atomicMax(reinterpret_cast<unsigned *>(addr), __float_as_uint(v));
}

__global__ void heights_kernel(const float *__restrict__ voxels, const int *__restrict__ coors4,
                               const int *m_dev, int m_bound, int max_points, int f, int batch,
                               int ny, int nx, int osf, float *__restrict__ mn,
                               float *__restrict__ mx) {
  int m = m_dev ? *m_dev : m_bound;
  if (m > m_bound) m = m_bound;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < m; v += gridDim.x * blockDim.x) {
    float lo = 100.0f, hi = -200.0f;
    for (int p = 0; p < max_points; ++p) {
      const float z = voxels[((size_t)v * max_points + p) * f + 2];
      const float zmin = (z == 0.0f) ? 100.0f : z;           // voxels_[voxels_==0] = 100   (:82)
      const float zmax = (zmin == 100.0f) ? -200.0f : z;     // voxels_[voxels_==100] = -200 (:84)
      lo = zmin < lo ? zmin : lo;
      hi = zmax > hi ? zmax : hi;
    }
    const int4 c = reinterpret_cast<const int4 *>(coors4)[v];
    const int cy = c.z / osf, cx = c.w / osf;
    if (c.x < 0 || c.x >= batch || cy < 0 || cy >= ny || cx < 0 || cx >= nx) continue;
    const size_t cell = ((size_t)c.x * ny + cy) * nx + cx;
    atomic_min_f32(&mn[cell], lo);
    atomic_max_f32(&mx[cell], hi);
  }
}

}  // namespace

extern "C" int sf_voxel_heights(const float *voxels, const int *coors4, const int *m_dev,
                                int m_bound, int max_points, int f, int batch, int ny, int nx,
                                int out_size_factor, float *min_height, float *max_height,
                                void *stream) {
  hipStream_t st = (hipStream_t)stream;
  SF_CHECK_ARG(m_bound >= 0 && max_points >= 1 && f >= 3 && batch >= 1 && ny >= 1 && nx >= 1 &&
                   out_size_factor >= 1,
               "bad shape (need >= 3 point features: z is column 2)");
  SF_CHECK_ARG(min_height && max_height, "output pointer is NULL");
  const long long cells = (long long)batch * ny * nx;
  hipLaunchKernelGGL(heights_init_kernel, dim3(sf_grid(cells, 256)), dim3(256), 0, st, min_height,
                     max_height, cells);
  SF_CHECK_LAUNCH();
  if (m_bound == 0) return SF_OK;
  SF_CHECK_ARG(voxels && coors4, "voxels / coors is NULL");
  hipLaunchKernelGGL(heights_kernel, dim3(sf_grid(m_bound, 256)), dim3(256), 0, st, voxels, coors4,
                     m_dev, m_bound, max_points, f, batch, ny, nx, out_size_factor, min_height,
                     max_height);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// SparseConvTensor.dense() + view(N, C*D, H, W) for gfx950
// (mmdet3d/ops/spconv/structure.py:55-64, mmdet3d/models/middle_encoders/sparse_encoder.py:122-125).
// out[b][c*D + z][y][x] = features[row][c]; everything else zero.
//
// HBM-bound: B*C*D*H*W*4 bytes written once.  A row-major scatter (one 4-byte store per (row,
// channel), each into a different channel plane) made the L2 fetch every line it touched: measured
// 180 MB fetched to write 33 MB.  Instead the map is written output-stationary:
//   1. the c = 0 planes of `out` (contiguous, D*H*W floats per sample) serve as a temporary cell ->
//      row map: zero them (a KERNEL, not hipMemsetAsync: a memset node between the kernel nodes of
//      a captured linear graph was not ordered with them on ROCm 7.2 -- replays left marks of the
//      previous frame in the map, tools/split_check.py), then every row writes row+1 (as bits) at
//      its own cell;
//   2. a workgroup owns 64 consecutive x of one (b, z, y) line: it reads the 64 map entries, stages
//      the active rows through LDS (coalesced 16-byte row reads) and writes every channel plane
//      with full 256-byte wave stores -- zeros included, so there is no separate zero fill, and
//      the map entries are overwritten by the real channel-0 values last.
#include "sf_common.h"

namespace {

__global__ void dense_zero_kernel(float *__restrict__ out, int C, int B, int D, size_t plane) {
  const size_t per = (size_t)D * plane, total = per * B;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / per;
    out[b * C * per + (i - b * per)] = 0.0f;
  }
}

__global__ void dense_mark_kernel(const int *__restrict__ idx, const int *n_dev, int n_bound, int C,
                                  int B, int D, int H, int W, float *__restrict__ out) {
  int n = n_dev ? *n_dev : n_bound;
  if (n > n_bound) n = n_bound;
  const size_t plane = (size_t)H * W;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    int4 p = reinterpret_cast<const int4 *>(idx)[r];
    if (p.x < 0 || p.x >= B || p.y < 0 || p.y >= D || p.z < 0 || p.z >= H || p.w < 0 || p.w >= W)
      continue;
    reinterpret_cast<int *>(out)[((size_t)p.x * C * D + p.y) * plane + (size_t)p.z * W + p.w] = r + 1;
  }
}

template <int CT>  // channels staged per pass (C is processed in passes of CT)
__global__ void __launch_bounds__(256)
dense_write_kernel(const float *__restrict__ feat, const int *n_dev, int n_bound, int C, int B, int D,
                   int H, int W, float *__restrict__ out) {
  __shared__ float tile[64][CT + 1];
  __shared__ int rows[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int n = n_dev ? *n_dev : n_bound;
  if (n > n_bound) n = n_bound;
  const int xt = (W + 63) / 64;
  const size_t plane = (size_t)H * W;
  const long long lines = (long long)B * D * H * xt;
  for (long long li = blockIdx.x; li < lines; li += gridDim.x) {
    const int x0 = (int)(li % xt) * 64;
    long long rest = li / xt;
    const int y = (int)(rest % H);
    rest /= H;
    const int z = (int)(rest % D), b = (int)(rest / D);
    const size_t cell = (size_t)y * W + x0;
    float *__restrict__ ob = out + ((size_t)b * C * D + z) * plane + cell;  // channel 0 of the line
    __syncthreads();  // previous line's tile / rows are consumed
    if (tid < 64) {
      // a map entry is row + 1 of THIS frame or 0; anything else (it cannot be, but a stray value
      // would index far outside `feat`) counts as empty
      const int r = (x0 + tid < W) ? reinterpret_cast<const int *>(ob)[tid] - 1 : -1;
      rows[tid] = (r >= 0 && r < n) ? r : -1;
    }
    for (int c0 = C - (C % CT ? C % CT : CT); c0 >= 0; c0 -= CT) {  // channel 0 (the map) goes last
      const int cn = (C - c0 < CT) ? C - c0 : CT;
      __syncthreads();
      // stage: row-major reads (consecutive threads = consecutive channels of one row)
      for (int i = tid; i < 64 * cn; i += 256) {
        const int rl = i / cn, c = i - rl * cn;
        const int r = rows[rl];
        tile[rl][c] = r >= 0 ? feat[(size_t)r * C + c0 + c] : 0.0f;
      }
      __syncthreads();
      // write: lane = x, one channel plane per wave iteration, descending so that c = 0 is last
      for (int c = cn - 1 - wave; c >= 0; c -= 4)
        if (x0 + lane < W) ob[(size_t)(c0 + c) * D * plane + lane] = tile[lane][c];
    }
  }
}

}  // namespace

namespace {

int dense_check(const void *features, const void *indices, int n_bound, int channels, int batch, int D,
                int H, int W, const void *out, bool need_feat, bool need_idx) {
  SF_CHECK_ARG(channels >= 1 && batch >= 1 && D >= 1 && H >= 1 && W >= 1 && n_bound >= 0,
               "bad dense shape");
  SF_CHECK_ARG(out, "out is NULL");
  SF_CHECK_ARG(n_bound == 0 || ((!need_feat || features) && (!need_idx || indices)),
               "features / indices is NULL");
  return SF_OK;
}

}  // namespace

// first half of sf_dense_bev: the cell -> row map in the c = 0 planes of `out` (needs the output
// coordinates only, so a pipeline can run it beside the convolutions)
extern "C" int sf_dense_bev_mark(const int *indices, const int *n_dev, int n_bound, int channels,
                                 int batch, int D, int H, int W, float *out, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  int rc = dense_check(nullptr, indices, n_bound, channels, batch, D, H, W, out, false, true);
  if (rc != SF_OK) return rc;
  const size_t plane = (size_t)H * W;
  hipLaunchKernelGGL(dense_zero_kernel, dim3(sf_grid((long long)batch * D * plane, 256)), dim3(256), 0,
                     st, out, channels, batch, D, plane);
  SF_CHECK_LAUNCH();
  if (n_bound > 0) {
    hipLaunchKernelGGL(dense_mark_kernel, dim3(sf_grid(n_bound, 256)), dim3(256), 0, st, indices,
                       n_dev, n_bound, channels, batch, D, H, W, out);
    SF_CHECK_LAUNCH();
  }
  return SF_OK;
}

// second half: every channel plane written from the rows the map names
extern "C" int sf_dense_bev_write(const float *features, const int *n_dev, int n_bound, int channels,
                                  int batch, int D, int H, int W, float *out, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  int rc = dense_check(features, nullptr, n_bound, channels, batch, D, H, W, out, true, false);
  if (rc != SF_OK) return rc;
  const long long lines = (long long)batch * D * H * ((W + 63) / 64);
  const int grid = (int)(lines < 8192 ? lines : 8192);
  hipLaunchKernelGGL((dense_write_kernel<128>), dim3(grid), dim3(256), 0, st, features, n_dev, n_bound,
                     channels, batch, D, H, W, out);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_dense_bev(const float *features, const int *indices, const int *n_dev,
                            int n_bound, int channels, int batch, int D, int H, int W, float *out,
                            void *stream) {
  int rc = dense_check(features, indices, n_bound, channels, batch, D, H, W, out, true, true);
  if (rc != SF_OK) return rc;
  rc = sf_dense_bev_mark(indices, n_dev, n_bound, channels, batch, D, H, W, out, stream);
  if (rc != SF_OK) return rc;
  return sf_dense_bev_write(features, n_dev, n_bound, channels, batch, D, H, W, out, stream);
}

// SparseConvTensor.dense() + view(N, C*D, H, W) for gfx950
// (mmdet3d/ops/spconv/structure.py:55-64, mmdet3d/models/middle_encoders/sparse_encoder.py:122-125).
// out[b][c*D + z][y][x] = features[row][c]; everything else zero.  HBM-bound: B*C*D*H*W*4 bytes of
// zero fill plus one scattered 4-byte store per (row, channel).
#include "sf_common.h"

namespace {

__global__ void dense_scatter_kernel(const float *__restrict__ feat, const int *__restrict__ idx,
                                     const int *n_dev, int n_bound, int C, int B, int D, int H,
                                     int W, float *__restrict__ out) {
  int n = n_dev ? *n_dev : n_bound;
  if (n > n_bound) n = n_bound;
  const long long total = (long long)n * C;
  const size_t plane = (size_t)H * W;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    // consecutive threads walk the rows for one channel: neighbouring rows are usually
    // neighbouring cells, so stores of a wave fall into few cache lines
    int c = (int)(e / n), r = (int)(e - (long long)c * n);
    int4 p = reinterpret_cast<const int4 *>(idx)[r];
    if (p.x < 0 || p.x >= B || p.y < 0 || p.y >= D || p.z < 0 || p.z >= H || p.w < 0 || p.w >= W)
      continue;
    out[(((size_t)p.x * C + c) * D + p.y) * plane + (size_t)p.z * W + p.w] = feat[(size_t)r * C + c];
  }
}

}  // namespace

extern "C" int sf_dense_bev(const float *features, const int *indices, const int *n_dev,
                            int n_bound, int channels, int batch, int D, int H, int W, float *out,
                            void *stream) {
  hipStream_t st = (hipStream_t)stream;
  SF_CHECK_ARG(channels >= 1 && batch >= 1 && D >= 1 && H >= 1 && W >= 1 && n_bound >= 0,
               "bad dense shape");
  SF_CHECK_ARG(out, "out is NULL");
  size_t bytes = (size_t)batch * channels * D * H * W * sizeof(float);
  SF_CHECK_HIP(hipMemsetAsync(out, 0, bytes, st));
  if (n_bound == 0) return SF_OK;
  SF_CHECK_ARG(features && indices, "features / indices is NULL");
  hipLaunchKernelGGL(dense_scatter_kernel, dim3(sf_grid((long long)n_bound * channels, 256)),
                     dim3(256), 0, st, features, indices, n_dev, n_bound, channels, batch, D, H, W,
                     out);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// Micro-benchmark: cycles per v_mfma_f32_16x16x4_f32 as a function of the number of independent
// accumulator chains a wave interleaves, with 1 or 2 waves per SIMD.  (Design input for the fp32
// conv kernels: how many chains does one wave need to keep the matrix pipe busy?)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ void k(float *out, int iters, unsigned long long *cyc) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CH>
void run(int waves_per_simd) {
  float *out; unsigned long long *cyc, h;
  int threads = 256 * waves_per_simd;  // 4 SIMDs
  hipMalloc(&out, 1024 * threads * 4); hipMalloc(&cyc, 8);
  int iters = 2000;
  hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
  hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  double per = (double)h / (iters * 8.0 * CH);
  printf("chains %d waves/SIMD %d: %.1f cycles per MFMA per wave -> pipe busy %.0f%%\n", CH, waves_per_simd, per,
         100.0 * 32.0 * waves_per_simd / per);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w = 1; w <= 2; ++w) { run<1>(w); run<2>(w); run<3>(w); run<4>(w); run<8>(w); }
  return 0;
}

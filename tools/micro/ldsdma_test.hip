#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// each lane gathers one 16-byte chunk of row idx[lane/4] (rows of 64 B) straight into LDS
__global__ void k(const float *feat, unsigned bytes, const int *idx, float *out) {
  __shared__ __attribute__((aligned(16))) float lds[2][256];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(feat), 0, bytes, 0x00020000);
  const int lane = threadIdx.x;
  for (int b = 0; b < 2; ++b) {
    int row = idx[b * 16 + lane / 4];
    unsigned off = (unsigned)row * 64u + (lane & 3) * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)&lds[b][0], 16, off, 0, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) etc.
  __syncthreads();
  for (int b = 0; b < 2; ++b)
    for (int i = lane; i < 256; i += 64) out[b * 256 + i] = lds[b][i];
}
int main() {
  float h[64 * 16]; for (int i = 0; i < 1024; ++i) h[i] = i;
  int hi[32]; for (int i = 0; i < 32; ++i) hi[i] = (i % 5 == 0) ? -1 : (i * 7) % 64;
  float *d, *o; int *di;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 512 * 4); hipMalloc(&di, sizeof(hi));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice); hipMemcpy(di, hi, sizeof(hi), hipMemcpyHostToDevice);
  hipMemset(o, 0xFF, 512 * 4);
  k<<<1, 64>>>(d, sizeof(h), di, o);
  float ho[512]; hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
    int row = hi[b * 16 + r]; float want = row < 0 ? 0.f : h[row * 16 + c];
    if (ho[b * 256 + r * 16 + c] != want) { if (bad < 5) printf("b%d r%d c%d got %f want %f\n", b, r, c, ho[b*256+r*16+c], want); ++bad; }
  }
  printf("bad %d\n", bad);
  return bad != 0;
}

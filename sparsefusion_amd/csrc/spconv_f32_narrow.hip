// Narrow-channel 3x3x3 sparse convolution (fp32, exact-f32 MFMA) for gfx950: Cin <= 32, Cout <= 32.
//
// Reference: indiceConv<float> (mmdet3d/ops/spconv/include/spconv/spconv_ops.h:260-361) + the
// BatchNorm / ReLU / residual modules around it (mmdet3d/ops/sparse_block.py:116-204).
//
// These layers (5->16, 16->16, 16->32, 32->32 of the SparseEncoder) move 64-128 bytes per gathered
// row and a few hundred flops per pair: they are bound by how many gathered rows a CU keeps in
// flight, not by the matrix pipe.  So the kernel is built around memory-level parallelism:
//
//   * a wave owns 16 output rows and keeps their sums in the MFMA accumulators for the whole
//     layer (no LDS tile, no atomics, no barrier after the filters are staged);
//   * the wave first loads ALL 27 neighbour indices of its rows (one dependent round trip instead
//     of 27), then streams the input rows with D offsets in flight at all times.  Gathers are raw
//     buffer loads whose offset for a missing neighbour lies beyond the descriptor's range: such a
//     load returns zeros without touching memory and without a branch, so the load stream is
//     straight-line code and the compiler's outstanding-load counts stay exact;
//   * the 27 filter matrices live in LDS in MFMA fragment order (staged once per workgroup,
//     workgroups are persistent over many 16-row tiles), read with immediate-offset ds_read_b128;
//   * offsets for which none of the 16 rows has a neighbour skip their MFMAs (wave-uniform
//     branch on a ballot);
//   * the product is taken transposed (filters = MFMA operand A, rows = operand B): a lane ends
//     with 4 consecutive channels of one row -> 16-byte stores, scale / shift / residual as
//     16-byte loads;
//   * tiles are dealt to workgroups in contiguous runs per XCD (blocks b and b + 8 share an
//     XCD's L2), so an XCD's L2 holds one eighth of the feature table plus a halo.
//
// Summation order per output row: centre offset first for SubM, then offsets ascending, channel
// groups ascending -- fixed; results do not depend on the launch shape.
#include <stdlib.h>

#include "spconv_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kK = 27;
constexpr unsigned kOob = 0x80000000u;   // beyond num_records of the feature descriptor

// visit order of the offsets: SubM adds the centre (offset 13) first (spconv_ops.h:300-303)
template <bool SUBM>
__device__ __host__ constexpr int visit(int kk) {
  return !SUBM ? kk : (kk == 0 ? 13 : (kk <= 13 ? kk - 1 : kk));
}

template <int CIN, int COUT, bool SUBM, int D, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
narrow_conv_f32_kernel(const float *__restrict__ feat, unsigned feat_bytes, const int *__restrict__ nbr, int nbr_stride,
                       const float *__restrict__ wpk, const float *__restrict__ scale,
                       const float *__restrict__ shift, const float *__restrict__ residual, int relu,
                       float *__restrict__ out, const int *n_out_dev, int n_out_bound) {
  constexpr int CINP = CIN < 16 ? 16 : CIN;
  constexpr int T = CINP / 16, CH = CIN / 4, NB = COUT / 16;
  constexpr int WFRAGS = kK * NB * T * 64;            // float4 fragments of the whole filter bank
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x4 *s_w = reinterpret_cast<f32x4 *>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, ln = lane & 15;
  {
    const f32x4 *__restrict__ wp4 = reinterpret_cast<const f32x4 *>(wpk);
    for (int i = tid; i < WFRAGS; i += WAVES * 64) s_w[i] = wp4[i];
  }
  __syncthreads();

  int n_out = n_out_dev ? *n_out_dev : n_out_bound;
  n_out = n_out < n_out_bound ? n_out : n_out_bound;
  const int tiles = (n_out + 15) >> 4;
  // contiguous runs of tiles per XCD, dealt to that XCD's waves round-robin
  const int nslots = gridDim.x >> 3;                   // workgroups per XCD (grid is a multiple of 8)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tpx = (tiles + 7) >> 3;
  const int lo = xcd * tpx;
  const int hi = (lo + tpx < tiles) ? lo + tpx : tiles;

  // feature table as a raw buffer of its real size (< 2 GiB, checked by the entry point): valid
  // rows are in range, kOob is not
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(feat), 0, feat_bytes, 0x00020000);

  for (int tile = lo + slot * WAVES + wave; tile < hi; tile += nslots * WAVES) {
    const int row = tile * 16 + ln;
    const bool live = row < n_out;
    const int rc = live ? row : n_out - 1;
    // ---- all 27 neighbour indices, one round trip -------------------------------------------
    unsigned voff[kK];
#pragma unroll
    for (int kk = 0; kk < kK; ++kk)
      voff[kk] = (unsigned)nbr[(size_t)visit<SUBM>(kk) * nbr_stride + rc];
#pragma unroll
    for (int kk = 0; kk < kK; ++kk) {
      const int ix = (int)voff[kk];
      voff[kk] = (live && ix >= 0 && g < CH) ? (unsigned)ix * (CIN * 4) + 16u * g : kOob;
    }
    // ---- stream the rows, D offsets in flight -------------------------------------------------
    f32x4 a[kK][T];
    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < kK + D; ++kk) {
      if (kk < kK) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
          auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[kk] + 64u * t, 0, 0);
          a[kk][t] = __builtin_bit_cast(f32x4, v);
        }
      }
      const int i = kk - D;
      if (i >= 0) {
        if (__builtin_amdgcn_ballot_w64(voff[i] != kOob) != 0) {
          const int k = visit<SUBM>(i);
#pragma unroll
          for (int t = 0; t < T; ++t) {
            f32x4 w[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) w[nb] = s_w[((k * NB + nb) * T + t) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nb][j], a[i][t][j], acc[nb], 0, 0, 0);
          }
        }
      }
    }
    // ---- epilogue: lane (g, ln) holds channels 16 nb + 4 g .. + 3 of row ln -----------------
    if (live) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int c = 16 * nb + 4 * g;
        f32x4 v = acc[nb];
        if (scale) {
          const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + c);
          const f32x4 sh = *reinterpret_cast<const f32x4 *>(shift + c);
          v = v * sc + sh;
        }
        if (residual) v += *reinterpret_cast<const f32x4 *>(residual + (size_t)row * COUT + c);
        if (relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
        }
        *reinterpret_cast<f32x4 *>(out + (size_t)row * COUT + c) = v;
      }
    }
  }
}

template <int CIN, int COUT, bool SUBM, int D, int WAVES>
int launch_narrow(const SfConvArgs &a) {
  constexpr int CINP = CIN < 16 ? 16 : CIN;
  constexpr size_t lds = (size_t)kK * (COUT / 16) * (CINP / 16) * 64 * 16;
  static_assert(lds <= 160 * 1024, "filter bank does not fit LDS");
  auto kern = narrow_conv_f32_kernel<CIN, COUT, SUBM, D, WAVES>;
  static SfSlotCache cache = {};  // resident workgroups per device
  int slots = sf_resident_slots(cache, kern, WAVES * 64, lds);
  SF_CHECK_ARG(slots > 0, "cannot query the device for the resident workgroups");
  slots &= ~7;                    // a multiple of 8: workgroups are dealt to the XCDs
  if (slots < 8) slots = 8;
  const int tiles = sf_div_up(a.n_out_bound, 16);
  int grid = (sf_div_up(tiles, WAVES) + 7) & ~7;     // one tile per wave is the least useful grid
  if (grid > slots) grid = slots;
  sf_note_kernel("narrow_conv_f32_kernel<%d, %d, %s, %d, %d>", CIN, COUT, SUBM ? "true" : "false", D, WAVES);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, a.st, a.feat, a.feat_bytes, a.nbr, a.nbr_stride, a.wpk,
                     a.scale, a.shift, a.residual, a.relu, a.out, a.n_out_dev, a.n_out_bound);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

}  // namespace

// Returns SF_OK after a launch, 1 when the shape is not one of this kernel's (the caller then
// uses the general kernels), or a negative SF_ERR_* code.
int sf_narrow_conv_f32(int cin_padded, int cout, const SfConvArgs &a) {
  if (a.K != kK) return 1;
  const bool subm = a.k_first == 13;
  if (!subm && a.k_first >= 0) return 1;
#define SF_NARROW(CI, CO, DEPTH, WV)                                          \
  if (cin_padded == CI && cout == CO)                                         \
    return subm ? launch_narrow<CI, CO, true, DEPTH, WV>(a)                   \
                : launch_narrow<CI, CO, false, DEPTH, WV>(a);
  SF_NARROW(8, 16, 9, 4)
  SF_NARROW(16, 16, 9, 4)
  SF_NARROW(16, 32, 9, 4)
  SF_NARROW(8, 32, 9, 4)
  // 32 -> 32: the filter bank is 108 KB, one 16-wave workgroup per CU shares it
  SF_NARROW(32, 32, 5, 16)
#undef SF_NARROW
  return 1;
}

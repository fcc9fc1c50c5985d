// Output-stationary fp32 sparse convolution over MASK-ORDERED rows (gfx950).
//
// Reference: indiceConv<float> (mmdet3d/ops/spconv/include/spconv/spconv_ops.h:260-361).  Same
// arithmetic as spconv_f32.hip (exact fp32 FMA chains on v_mfma_f32_16x16x4_f32), different work
// decomposition.  The tiled kernels there compact the (input, output) pairs of every offset so
// that no empty slot is multiplied, and pay for it with ~300 bookkeeping instructions per phase
// and wave and a read-add-write of an LDS output tile (matrix pipe busy 56 %).  Here nothing is
// compacted and the sums never leave the MFMA accumulators; instead the ROWS are visited in the
// order of their 27-bit neighbour mask (which offsets have a partner), so that the rows of a tile
// want the same offsets and whole (offset, 16-row block) products can be skipped:
//   measured on the synthetic nuScenes frame, MFMA rows issued / pairs = 1.11-1.21 with 16..64-row
//   blocks in mask order, against 1.46 (stage 4) .. 4.54 (stage 1) for all 27 offsets in natural
//   order and ~1.18 for compaction with 16-row padding.
// LiDAR surfaces make the masks cluster (ground voxels have their partners in the same z layer).
//
//   * sf_row_order_by_mask: mask per output row from the neighbour table, stable radix sort of
//     (mask, row) -> row_order; rows beyond *n_dev sort to the end.  One call per rulebook, shared
//     by every layer that uses the rulebook; no host synchronisation.
//   * kernel: a workgroup owns TM consecutive entries of row_order.  Setup: the tile's columns of
//     the neighbour table -> LDS, one ballot per offset gives the 16-row blocks that have a
//     partner, wave 0 lists the active offsets (SubM centre first, then ascending: the reference's
//     summation order, spconv_ops.h:300-356).  Per active offset (a phase, one barrier each): the
//     tile's input rows are gathered two phases ahead with raw buffer loads (an empty slot is an
//     out-of-range offset: zeros, no memory access, no branch), parked in one of two XOR-swizzled
//     LDS buffers one phase ahead, and multiplied from LDS by the filters, which stream from L2
//     into registers one phase ahead in the fragment order sf_pack_filters_f32 wrote.  Waves
//     form a CG x RG grid (column groups x row-block groups).  The product is transposed
//     (out^T = W^T x^T): a lane ends with 4 consecutive channels of one output row -> 16-byte
//     stores straight from the accumulators after the folded BN / residual / ReLU epilogue.
// Per output element the FMA chain is: centre offset, then offsets ascending, channels ascending
// -- fixed, no atomics, independent of the row order.
#include <hipcub/hipcub.hpp>

#include "sf_common.h"

namespace {

constexpr int kMaxK = 27;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int load_n(const int *n_dev, int n_bound) {
  int n = n_dev ? *n_dev : n_bound;
  return n < n_bound ? n : n_bound;
}

template <int CIN>
struct ATileF {
  static constexpr int CH = CIN / 4;                  // 16-byte chunks per row
  static constexpr int R = (CH >= 16) ? 1 : 16 / CH;  // rows per 256-byte bank row
  static constexpr int M = (CH >= 16) ? 15 : CH - 1;
  // XOR swizzle of the chunk position: the 16 rows of one fragment read spread over the banks
  __device__ static __forceinline__ int pos(int p, int c) { return c ^ ((p / R) & M); }
};

__global__ void row_mask_kernel(const int *__restrict__ nbr, int nbr_stride, int K,
                                const int *n_dev, int n_bound, unsigned *__restrict__ keys,
                                int *__restrict__ rows) {
  const int n = load_n(n_dev, n_bound);
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n_bound; o += gridDim.x * blockDim.x) {
    unsigned m = 0xFFFFFFFFu;  // rows past the count sort behind every real row
    if (o < n) {
      m = 0;
      for (int k = 0; k < K; ++k) m |= (unsigned)(nbr[(size_t)k * nbr_stride + o] >= 0) << k;
    }
    keys[o] = m;
    rows[o] = o;
  }
}

template <int CIN, int COUT, int TM, int CG, int WAVES, int MINB>
__global__ void __launch_bounds__(WAVES * 64, MINB)
conv_f32_ordered_kernel(const float *__restrict__ feat, unsigned feat_bytes,
                        const int *__restrict__ nbr, int nbr_stride, int K, int k_first,
                        const int *__restrict__ order, const f32x4 *__restrict__ wp,
                        const float *__restrict__ scale, const float *__restrict__ shift,
                        const float *__restrict__ residual, int relu, float *__restrict__ out,
                        const int *n_out_dev, int n_out_bound) {
  constexpr int NT = WAVES * 64, T = CIN / 16, ROWB = CIN * 4, CH = CIN / 4;
  constexpr int NB = COUT / 16, NBW = NB / CG, RG = WAVES / CG, MBT = TM / 16, MBW = MBT / RG;
  constexpr int ITEMS = TM * CH, GI = (ITEMS + NT - 1) / NT;
  static_assert(TM % 16 == 0 && TM <= 64 && NB % CG == 0 && WAVES % CG == 0 && MBT % RG == 0 &&
                    CIN % 16 == 0, "tile shape");

  // dynamic LDS (the 128-channel tile passes the 64 KiB static limit):
  // rows [2][TM*CH] x 16 B | neighbour columns [27][TM] | tile rows [TM] | block masks | list | count
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  u32x4(*s_a)[TM * CH] = reinterpret_cast<u32x4(*)[TM * CH]>(lds_raw);
  int *s_nbr = reinterpret_cast<int *>(lds_raw + (size_t)2 * TM * CH * 16);
  int *s_row = s_nbr + kMaxK * TM;
  int *s_kmask = s_row + TM;
  int *s_list = s_kmask + 32;
  int &s_nact = s_list[32];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % CG, rg = wave / CG;
  const int g = lane >> 4, ln = lane & 15;
  const int n_out = load_n(n_out_dev, n_out_bound);
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(feat), 0, feat_bytes, 0x00020000);
  const int kf = (k_first >= 0 && k_first < K) ? k_first : -1;
  const int tiles = (n_out + TM - 1) / TM;

  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int row0 = tile * TM;
    // ---- setup ------------------------------------------------------------------------
    if (tid < TM) {
      const int e = row0 + tid;
      s_row[tid] = e < n_out ? (order ? order[e] : e) : -1;
    }
    __syncthreads();
    for (int k = wave; k < K; k += WAVES) {
      const int o = lane < TM ? s_row[lane] : -1;
      const int v = o >= 0 ? nbr[(size_t)k * nbr_stride + o] : -1;
      if (lane < TM) s_nbr[k * TM + lane] = v;
      const unsigned long long bal = __ballot(v >= 0);
      int bm = 0;
#pragma unroll
      for (int b = 0; b < MBT; ++b) bm |= (int)(((bal >> (16 * b)) & 0xFFFFull) != 0) << b;
      if (lane == 0) s_kmask[k] = bm;
    }
    __syncthreads();
    if (wave == 0) {
      const int bm = lane < K ? s_kmask[lane] : 0;
      const unsigned long long act = __ballot(bm != 0);
      const int below = __popcll(act & ((1ull << lane) - 1ull));
      const bool cf = kf >= 0 && ((act >> kf) & 1ull);  // the centre offset goes first
      int pos = below;
      if (cf) pos = lane == kf ? 0 : (lane < kf ? below + 1 : below);
      if (bm != 0) s_list[pos] = lane | (bm << 8);
      if (lane == 0) s_nact = __popcll(act);
    }
    __syncthreads();
    const int nact = s_nact;

    f32x4 acc[MBW][NBW];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int i = 0; i < NBW; ++i) acc[mb][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 greg[2][GI];
    f32x4 bf[2][NBW][T];
#pragma unroll
    for (int m = 0; m < GI; ++m) greg[0][m] = greg[1][m] = (u32x4){0u, 0u, 0u, 0u};

    // one phase; PAR = parity of kk: LDS buffer / filter set of phase kk, register set that
    // receives the rows of phase kk+2
    auto phase = [&](int kk, auto PAR) {
      constexpr int par = decltype(PAR)::value;
      // (1) filters of phase kk+1 first, then rows of phase kk+2
      {
        int i1 = kk + 1 < 0 ? 0 : kk + 1;
        i1 = i1 < nact ? i1 : nact - 1;
        const int k1 = s_list[i1] & 31;
#pragma unroll
        for (int i = 0; i < NBW; ++i)
#pragma unroll
          for (int t = 0; t < T; ++t)
            bf[par ^ 1][i][t] = wp[(((size_t)k1 * NB + (cg * NBW + i)) * T + t) * 64 + lane];
        const int v2 = kk + 2;
        const int k2 = s_list[v2 < nact ? v2 : nact - 1] & 31;
#pragma unroll
        for (int m = 0; m < GI; ++m) {
          const int it = tid + m * NT, p = it / CH, c = it % CH;
          int src = s_nbr[k2 * TM + (p < TM ? p : 0)];
          src = (v2 < nact && it < ITEMS) ? src : -1;  // -1: out of range, zeros, no fetch
          greg[par][m] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)src * ROWB + c * 16, 0, 0);
        }
      }
      // (2) this phase: the row blocks that have a partner under this offset.  In mask order
      //     nearly every phase has all of the wave's blocks active: that case is a straight line
      //     (a branch per block and MFMA measured a wait in front of every MFMA); the rest takes
      //     the blocks one by one.
      if (kk >= 0 && kk < nact) {
        const int bmw =
            (__builtin_amdgcn_readfirstlane(s_list[kk]) >> (8 + rg * MBW)) & ((1 << MBW) - 1);
        const char *sa = reinterpret_cast<const char *>(s_a[par]);
        auto a_at = [&](int mb, int t) {
          const int p = (rg * MBW + mb) * 16 + ln;
          return *reinterpret_cast<const f32x4 *>(sa + p * ROWB + 16 * ATileF<CIN>::pos(p, 4 * t + g));
        };
        if (bmw == (1 << MBW) - 1) {
          f32x4 a[2][MBW];
#pragma unroll
          for (int mb = 0; mb < MBW; ++mb) a[0][mb] = a_at(mb, 0);
#pragma unroll
          for (int t = 0; t < T; ++t) {
            if (t + 1 < T) {
#pragma unroll
              for (int mb = 0; mb < MBW; ++mb) a[(t + 1) & 1][mb] = a_at(mb, t + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
                for (int i = 0; i < NBW; ++i)
                  acc[mb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[par][i][t][j], a[t & 1][mb][j],
                                                                   acc[mb][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else if (bmw != 0) {
#pragma unroll
          for (int mb = 0; mb < MBW; ++mb)
            if ((bmw >> mb) & 1) {  // uniform
#pragma unroll
              for (int t = 0; t < T; ++t) {
                const f32x4 av = a_at(mb, t);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                  for (int i = 0; i < NBW; ++i)
                    acc[mb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[par][i][t][j], av[j],
                                                                     acc[mb][i], 0, 0, 0);
              }
            }
        }
      }
      // (3) park the rows of phase kk+1 (issued one phase ago) in the other LDS buffer
      {
        char *sn = reinterpret_cast<char *>(s_a[par ^ 1]);
#pragma unroll
        for (int m = 0; m < GI; ++m) {
          const int it = tid + m * NT, p = it / CH, c = it % CH;
          if (ITEMS % NT == 0 || it < ITEMS)
            *reinterpret_cast<u32x4 *>(sn + p * ROWB + 16 * ATileF<CIN>::pos(p, c)) = greg[par ^ 1][m];
        }
      }
      __syncthreads();
    };
    if (nact > 0)
      for (int kk = -2; kk < nact; kk += 2) {
        phase(kk, std::integral_constant<int, 0>{});
        phase(kk + 1, std::integral_constant<int, 1>{});
      }

    // ---- epilogue: folded BN, residual, ReLU, straight from the accumulators ---------------
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
      const int row = s_row[(rg * MBW + mb) * 16 + ln];
      if (row < 0) continue;
#pragma unroll
      for (int i = 0; i < NBW; ++i) {
        const int c = (cg * NBW + i) * 16 + 4 * g;
        f32x4 v = acc[mb][i];
        if (scale) {
          v = v * *reinterpret_cast<const f32x4 *>(scale + c) +
              *reinterpret_cast<const f32x4 *>(shift + c);
        }
        if (residual) v += *reinterpret_cast<const f32x4 *>(residual + (size_t)row * COUT + c);
        if (relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
        }
        *reinterpret_cast<f32x4 *>(out + (size_t)row * COUT + c) = v;
      }
    }
    __syncthreads();  // s_row / lists are rewritten by the next tile
  }
}

struct OrdArgs {
  const float *feat;
  unsigned feat_bytes;
  const int *nbr;
  int nbr_stride, K, k_first;
  const int *order;
  const float *wpk, *scale, *shift, *residual;
  int relu;
  float *out;
  const int *n_out_dev;
  int n_out_bound;
  hipStream_t st;
};

template <int CIN, int COUT, int TM, int CG, int WAVES, int MINB>
int launch_ordered(const OrdArgs &a) {
  constexpr size_t lds = (size_t)2 * TM * (CIN / 4) * 16 + (size_t)(kMaxK * TM + TM + 32 + 33) * 4;
  static_assert(lds <= 160 * 1024, "tile does not fit LDS");
  auto kern = conv_f32_ordered_kernel<CIN, COUT, TM, CG, WAVES, MINB>;
  static bool attr_set = false;
  if (lds > 48 * 1024 && !attr_set) {
    SF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int tiles = sf_div_up(a.n_out_bound, TM);
  const int grid = tiles < 8192 ? tiles : 8192;
  hipLaunchKernelGGL(kern, dim3(grid),
                     dim3(WAVES * 64), lds, a.st, a.feat, a.feat_bytes, a.nbr, a.nbr_stride, a.K,
                     a.k_first, a.order, reinterpret_cast<const f32x4 *>(a.wpk), a.scale, a.shift,
                     a.residual, a.relu, a.out, a.n_out_dev, a.n_out_bound);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// Wave grid per shape: the column groups CG divide COUT/16; 8 waves where a wave's filter
// registers (2 x NBW x CIN/16 float4) would otherwise not fit.
template <int CIN>
int dispatch_ordered(int cout, const OrdArgs &a) {
  switch (cout) {
    case 16: return launch_ordered<CIN, 16, 64, 1, 4, 2>(a);
    case 32: return launch_ordered<CIN, 32, 64, 2, 4, 2>(a);
    case 64: return launch_ordered<CIN, 64, 64, 4, 4, (CIN <= 64 ? 3 : 2)>(a);
    case 128:
      if constexpr (CIN >= 128) return launch_ordered<CIN, 128, 64, 8, 8, 1>(a);
      else return launch_ordered<CIN, 128, 64, 4, 4, 2>(a);
  }
  sf_set_error("unsupported cout=%d (16, 32, 64 or 128)", cout);
  return SF_ERR_BAD_ARG;
}

}  // namespace

extern "C" size_t sf_row_order_workspace(int n_bound) {
  if (n_bound < 1) n_bound = 1;
  size_t tmp = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const unsigned *)nullptr, (unsigned *)nullptr,
                                     (const int *)nullptr, (int *)nullptr, n_bound, 0, 32, 0);
  const size_t arr = ((size_t)n_bound * 4 + 255) & ~(size_t)255;
  return 3 * arr + ((tmp + 255) & ~(size_t)255);
}

extern "C" int sf_row_order_by_mask(const int *nbr, int nbr_stride, int K, const int *n_dev,
                                    int n_bound, int *row_order, void *workspace,
                                    size_t workspace_bytes, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  SF_CHECK_ARG(K >= 1 && K <= kMaxK, "kernel volume %d not in [1,%d]", K, kMaxK);
  SF_CHECK_ARG(n_bound >= 0 && nbr_stride >= n_bound, "n_bound=%d nbr_stride=%d", n_bound, nbr_stride);
  if (n_bound == 0) return SF_OK;
  SF_CHECK_ARG(nbr && row_order && workspace, "NULL pointer");
  if (workspace_bytes < sf_row_order_workspace(n_bound)) {
    sf_set_error("row-order workspace: %zu bytes given, %zu needed", workspace_bytes,
                 sf_row_order_workspace(n_bound));
    return SF_ERR_WORKSPACE;
  }
  const size_t arr = ((size_t)n_bound * 4 + 255) & ~(size_t)255;
  char *ws = (char *)workspace;
  unsigned *keys = (unsigned *)ws, *keys_out = (unsigned *)(ws + arr);
  int *rows = (int *)(ws + 2 * arr);
  void *tmp = ws + 3 * arr;
  size_t tmp_bytes = workspace_bytes - 3 * arr;
  hipLaunchKernelGGL(row_mask_kernel, dim3(sf_grid(n_bound, 256)), dim3(256), 0, st, nbr, nbr_stride,
                     K, n_dev, n_bound, keys, rows);
  SF_CHECK_LAUNCH();
  SF_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys_out, rows, row_order,
                                                  n_bound, 0, 32, st));
  return SF_OK;
}

extern "C" int sf_sparse_conv_f32_ordered(const float *features_in, int n_in_bound, int cin_padded,
                                          const int *nbr, int nbr_stride, int K, int k_first,
                                          const int *row_order, const float *w_packed, int cout,
                                          const float *scale, const float *shift,
                                          const float *residual, int relu, float *out,
                                          const int *n_out_dev, int n_out_bound, void *stream) {
  SF_CHECK_ARG(K >= 1 && K <= kMaxK, "kernel volume %d not in [1,%d]", K, kMaxK);
  SF_CHECK_ARG(n_out_bound >= 0 && nbr_stride >= n_out_bound, "n_out_bound=%d nbr_stride=%d",
               n_out_bound, nbr_stride);
  SF_CHECK_ARG((scale == nullptr) == (shift == nullptr), "scale and shift go together");
  if (n_out_bound == 0) return SF_OK;
  SF_CHECK_ARG(features_in && nbr && w_packed && out, "NULL tensor pointer");
  SF_CHECK_ARG(((uintptr_t)features_in & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
                   ((uintptr_t)w_packed & 15) == 0 && ((uintptr_t)residual & 15) == 0 &&
                   ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0,
               "feature / filter / epilogue pointers must be 16-byte aligned");
  const long long bytes = (long long)n_in_bound * cin_padded * 4;
  SF_CHECK_ARG(n_in_bound >= 1 && bytes < (1ll << 32) - 4096,
               "the input feature table (%d rows) must stay below 4 GiB", n_in_bound);
  OrdArgs a{features_in, (unsigned)bytes, nbr, nbr_stride, K, k_first, row_order, w_packed, scale,
            shift, residual, relu, out, n_out_dev, n_out_bound, (hipStream_t)stream};
  switch (cin_padded) {
    case 16: return dispatch_ordered<16>(cout, a);
    case 32: return dispatch_ordered<32>(cout, a);
    case 64: return dispatch_ordered<64>(cout, a);
    case 128: return dispatch_ordered<128>(cout, a);
  }
  sf_set_error("unsupported cin_padded=%d for the ordered kernel (16, 32, 64 or 128)", cin_padded);
  return SF_ERR_BAD_ARG;
}

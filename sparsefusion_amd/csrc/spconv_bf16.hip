// Output-stationary sparse convolution, bf16 storage / bf16 MFMA / fp32 accumulation (gfx950).
//
// Reference: indiceConv (mmdet3d/ops/spconv/include/spconv/spconv_ops.h:260-361), which the
// reference runs in half precision when SparseEncoder.fp16_enabled is set
// (mmdet3d/models/middle_encoders/sparse_encoder.py:53,95 auto_fp16).  BASELINE.json configs[2] asks
// for the bf16 form on MI355X.
//
// Why this is a different kernel from spconv_f32.hip and not a retyped copy: the bf16 matrix
// pipe is 16x the fp32 one, so the per-offset contraction stops being the bound and the layer
// becomes a gather problem (HBM / L2 bytes and load latency).  Consequences:
//   * no compaction of (input,output) pairs: every offset is multiplied densely over the tile's
//     rows; an empty (row, offset) slot costs a zero fragment, not a fetch -- the gather is a raw
//     buffer load whose offset for an empty slot is out of range, which returns zeros WITHOUT a
//     memory access (and without a branch, so the loads of later offsets stay in flight);
//   * the sums stay in the MFMA accumulators for the whole layer (no LDS output tile, no
//     read-add-write), the product is computed transposed (out^T = W^T x^T) so that a lane ends up
//     with 4 consecutive channels of one output row and stores them with one 8-byte write;
//   * narrow layers (Cin <= 32, 27 offsets) use a wave-autonomous kernel without LDS or barriers;
//     wide layers stage the gathered rows once per workgroup in LDS (XOR-swizzled, double
//     buffered, one barrier per offset) and the waves split the output columns.
// Summation order per output element: centre offset first (SubM), then offsets ascending,
// channels ascending inside an MFMA -- fixed, no atomics.
//
// Algorithmic HBM bytes per layer: P*Cin*2 (gather) + N_out*Cout*2*(1+residual) + P*8 (indices).
#include <stdlib.h>

#include <type_traits>

#include "sf_common.h"

namespace {

constexpr int kMaxK = 27;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int load_n(const int *n_dev, int n_bound) {
  int n = n_dev ? *n_dev : n_bound;
  return n < n_bound ? n : n_bound;
}

// Per-lane MFMA operand of one k-step.  Cin >= 32: v_mfma_f32_16x16x32_bf16, lane (g = lane>>4,
// ln = lane&15) holds k = 8g..8g+7 of row/column ln (16 bytes).  Cin == 16: v_mfma_f32_16x16x16_bf16,
// k = 4g..4g+3 (8 bytes).
template <int CIN>
struct BFrag {
  using type = bf16x8;
  static constexpr int T = CIN / 32, E = 8;
  __device__ static __forceinline__ type gload(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(type, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  }
  __device__ static __forceinline__ f32x4 mma(type w, type x, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
  }
};
template <>
struct BFrag<16> {
  using type = bf16x4;
  static constexpr int T = 1, E = 4;
  __device__ static __forceinline__ type gload(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(type, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
  }
  __device__ static __forceinline__ f32x4 mma(type w, type x, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, w),
                                                     __builtin_bit_cast(s16x4, x), c, 0, 0, 0);
  }
};

// visit order of the offsets: kf (SubM centre) first, then ascending
__device__ __forceinline__ int visit_k(int i, int kf) {
  return kf < 0 ? i : (i == 0 ? kf : (i <= kf ? i - 1 : i));
}

// Epilogue of one accumulator: lane (g, ln) holds out[row][ch0 .. ch0+3] (transposed product).
template <int COUT>
__device__ __forceinline__ void store_acc(f32x4 v, int row, int ch0, const float *__restrict__ scale,
                                          const float *__restrict__ shift,
                                          const __bf16 *__restrict__ residual, int relu, void *out,
                                          int out_f32) {
  if (scale) {
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + ch0);
    const f32x4 sh = *reinterpret_cast<const f32x4 *>(shift + ch0);
    v = v * sc + sh;
  }
  const size_t e = (size_t)row * COUT + ch0;
  if (residual) {
    const bf16x4 r = *reinterpret_cast<const bf16x4 *>(residual + e);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
  }
  if (relu) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
  }
  if (out_f32) {
    *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(out) + e) = v;
  } else {
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];  // v_cvt_pk_bf16_f32: round to nearest even
    *reinterpret_cast<bf16x4 *>(reinterpret_cast<__bf16 *>(out) + e) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// Direct kernel (27 offsets, Cin <= 32): a wave owns 16*MB output rows and all output columns.
// Per offset every lane loads its own operand piece straight from the input row the neighbour
// table names; filter fragments come from L1/L2 (27 small matrices shared by every wave).
// Neighbour indices run D0 + D1 offsets ahead of the MFMAs, rows and filters D1 offsets ahead.
template <int CIN, int COUT, int MB, int D1>
__global__ void __launch_bounds__(256)
conv_bf16_direct_kernel(const void *__restrict__ feat, unsigned feat_bytes,
                        const int *__restrict__ nbr, int nbr_stride, int k_first,
                        const void *__restrict__ wpk, const float *__restrict__ scale,
                        const float *__restrict__ shift, const __bf16 *__restrict__ residual,
                        int relu, void *__restrict__ out, int out_f32, const int *n_out_dev,
                        int n_out_bound) {
  using FR = BFrag<CIN>;
  using frag = typename FR::type;
  constexpr int K = 27, D0 = 2, T = FR::T, E = FR::E;
  constexpr int NB = COUT / 16, ROWS = 16 * MB, ROWB = CIN * 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, ln = lane & 15;
  const int n_out = load_n(n_out_dev, n_out_bound);
  // Row blocks are dealt to the 8 XCDs in contiguous runs (workgroup b runs on XCD b % 8), so
  // every XCD's L2 holds one eighth of the feature table instead of all of it (at batch 8 the
  // table no longer fits one 4 MB L2).
  const int nblocks = (n_out + ROWS * 4 - 1) / (ROWS * 4), per = (nblocks + 7) >> 3;
  const int bj = blockIdx.x >> 3, blk = (blockIdx.x & 7) * per + bj;
  if (bj >= per || blk >= nblocks) return;
  const int row0 = (blk * 4 + wave) * ROWS;
  if (row0 >= n_out) return;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(feat), 0, feat_bytes, 0x00020000);
  const frag *__restrict__ wp = reinterpret_cast<const frag *>(wpk);
  const int kf = (k_first >= 0 && k_first < K) ? k_first : -1;

  f32x4 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // every array is indexed by compile-time constants only (the loop is fully unrolled)
  int idx[K][MB];
  frag a[K][MB][T];
  frag b[K][NB][T];
#pragma unroll
  for (int i = -(D0 + D1); i < K; ++i) {
    if (i + D0 + D1 < K) {
      const int kk = i + D0 + D1, k = visit_k(kk, kf);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int r = row0 + mb * 16 + ln;
        const int v = nbr[(size_t)k * nbr_stride + (r < n_out ? r : n_out - 1)];
        idx[kk][mb] = (r < n_out) ? v : -1;
      }
    }
    if (i + D1 >= 0 && i + D1 < K) {
      const int kk = i + D1, k = visit_k(kk, kf);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int t = 0; t < T; ++t) b[kk][nb][t] = wp[(((size_t)k * NB + nb) * T + t) * 64 + lane];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int t = 0; t < T; ++t)
          // idx == -1 wraps to an offset beyond feat_bytes: the load returns zeros, no fetch
          a[kk][mb][t] = FR::gload(rs, (unsigned)idx[kk][mb] * ROWB + t * 64 + g * (E * 2));
    }
    if (i >= 0) {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            acc[mb][nb] = FR::mma(b[i][nb][t], a[i][mb][t], acc[mb][nb]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }

#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int row = row0 + mb * 16 + ln;
    if (row < n_out) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        store_acc<COUT>(acc[mb][nb], row, nb * 16 + 4 * g, scale, shift, residual, relu, out,
                        out_f32);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Tile kernel (any Cin/Cout/K): a workgroup of WAVES waves owns TM output rows.  The waves form
// a CG x RG grid: CG groups split the output columns, RG groups split the tile's 16-row blocks.
// Per offset (a "phase") all threads gather the tile's input rows into registers two phases
// ahead and park them in one of two LDS buffers one phase ahead; a phase is
//     issue filters(kk+1), issue rows(kk+2) | MFMA(kk) from LDS | park rows(kk+1) | barrier.
template <int CIN>
struct ATileB {
  static constexpr int CH = CIN / 8;                    // 16-byte chunks per row
  static constexpr int R = (CH >= 16) ? 1 : 16 / CH;    // rows per 256-byte bank row
  static constexpr int M = (CH >= 16) ? 15 : CH - 1;
  // XOR swizzle of the chunk position: the 16 rows one fragment read touches spread over the
  // banks (ds_read_b128 is served per 16-lane group)
  __device__ static __forceinline__ int pos(int p, int c) { return c ^ ((p / R) & M); }
};

template <int CIN, int COUT, int TM, int CG, int WAVES, int MINB>
__global__ void __launch_bounds__(WAVES * 64, MINB)
conv_bf16_tile_kernel(const void *__restrict__ feat, unsigned feat_bytes,
                      const int *__restrict__ nbr, int nbr_stride, int K, int k_first,
                      const void *__restrict__ wpk, const float *__restrict__ scale,
                      const float *__restrict__ shift, const __bf16 *__restrict__ residual,
                      int relu, void *__restrict__ out, int out_f32, const int *n_out_dev,
                      int n_out_bound, int balance) {
  using FR = BFrag<CIN>;
  using frag = typename FR::type;
  constexpr int NT = WAVES * 64, T = FR::T, E = FR::E, ROWB = CIN * 2, CH = CIN / 8;
  constexpr int NB = COUT / 16, NBW = NB / CG, RG = WAVES / CG, MBT = TM / 16, MBW = MBT / RG;
  constexpr int ITEMS = TM * CH, GI = (ITEMS + NT - 1) / NT;
  static_assert(TM % 16 == 0 && NB % CG == 0 && WAVES % CG == 0 && MBT % RG == 0, "tile shape");

  __shared__ u32x4 s_a[2][TM * CH];
  __shared__ int s_nbr[kMaxK * TM];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % CG, rg = wave / CG;
  const int g = lane >> 4, ln = lane & 15;
  const int n_out = load_n(n_out_dev, n_out_bound);
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(feat), 0, feat_bytes, 0x00020000);
  const frag *__restrict__ wp = reinterpret_cast<const frag *>(wpk);
  const int kf = (k_first >= 0 && k_first < K) ? k_first : -1;

  // Tile height: TM, or (balance: the grid is then what is resident at once) the largest
  // tm <= TM that gives every workgroup the same number of tiles; tm need not be a multiple of
  // 16 -- the matrix pipe is not the bound, a partly empty last row block is free.  Measured:
  // plain TM-row tiles handed out by the hardware dispatcher are FASTER here (64x64: 55 vs 71 us,
  // 128x128: 60 vs 67-77 us): the layer is bound by load latency under a barrier per offset,
  // and workgroups that start at different times fill each other's stalls.
  // Tiles are dealt to the 8 XCDs in contiguous runs (workgroup b runs on XCD b % 8): nearby
  // output rows gather overlapping input rows, which then meet in one XCD's L2.
  const int G = gridDim.x;  // multiple of 8
  int tm = TM;
  if (balance) {
    const int rounds = (n_out + G * TM - 1) / (G * TM);
    tm = rounds > 0 ? (n_out + G * rounds - 1) / (G * rounds) : TM;
    tm = tm < 16 ? 16 : (tm > TM ? TM : tm);
  }
  const int tiles = (n_out + tm - 1) / tm;
  const int per = (tiles + 7) >> 3;
  const int nblk = (tm + 15) >> 4;  // 16-row blocks that hold rows

  for (int tix = blockIdx.x; tix < 8 * per; tix += G) {
    const int tile = (tix & 7) * per + (tix >> 3);
    if (tile >= tiles) continue;  // uniform for the workgroup
    const int row0 = tile * tm;
    const int row_end = (row0 + tm < n_out) ? row0 + tm : n_out;

    // ---- setup: the tile's columns of the neighbour table -> LDS
    for (int k = wave; k < K; k += WAVES) {
      for (int r = lane; r < TM; r += 64) {
        const int o = row0 + r;
        s_nbr[k * TM + r] = (o < row_end) ? nbr[(size_t)k * nbr_stride + o] : -1;
      }
    }
    __syncthreads();

    f32x4 acc[MBW][NBW];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int i = 0; i < NBW; ++i) acc[mb][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 greg[2][GI];
    frag bf[2][NBW][T];
#pragma unroll
    for (int m = 0; m < GI; ++m) greg[0][m] = greg[1][m] = (u32x4){0u, 0u, 0u, 0u};

    // one phase; PAR = parity of kk: LDS buffer / filter set of phase kk, register set that
    // receives the rows of phase kk+2
    auto phase = [&](int kk, auto PAR) {
      constexpr int par = decltype(PAR)::value;
      // (1) filters of phase kk+1 first, then rows of phase kk+2: the MFMAs below wait for the
      //     filters issued one phase ago, which must not sit behind this phase's row loads
      {
        const int k1c = visit_k(kk + 1 < K ? (kk + 1 < 0 ? 0 : kk + 1) : K - 1, kf);
#pragma unroll
        for (int i = 0; i < NBW; ++i)
#pragma unroll
          for (int t = 0; t < T; ++t)
            bf[par ^ 1][i][t] = wp[(((size_t)k1c * NB + (cg * NBW + i)) * T + t) * 64 + lane];
        const int v2 = kk + 2;
        const int k2 = visit_k(v2 < K ? v2 : K - 1, kf);
#pragma unroll
        for (int m = 0; m < GI; ++m) {
          const int it = tid + m * NT, p = it / CH, c = it % CH;
          int src = s_nbr[k2 * TM + (p < TM ? p : 0)];
          src = (v2 < K && it < ITEMS) ? src : -1;  // -1: offset out of range, zeros, no fetch
          greg[par][m] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)src * ROWB + c * 16, 0, 0);
        }
      }
      // (2) this phase: dense MFMA over the tile's row blocks.  The A fragments of channel
      //     group t+1 are read from LDS while group t multiplies (LDS latency under the MFMAs).
      if (kk >= 0 && kk < K) {
        const char *sa = reinterpret_cast<const char *>(s_a[par]);
        auto a_read = [&](frag(&a)[MBW], int t) {
#pragma unroll
          for (int mb = 0; mb < MBW; ++mb) {
            const int p = (rg * MBW + mb) * 16 + ln;
            if (rg * MBW + mb >= nblk) continue;  // uniform: block beyond the tile height
            if constexpr (E == 8) {
              a[mb] = *reinterpret_cast<const frag *>(sa + p * ROWB +
                                                      16 * ATileB<CIN>::pos(p, 4 * t + g));
            } else {
              a[mb] = *reinterpret_cast<const frag *>(sa + p * ROWB +
                                                      16 * ATileB<CIN>::pos(p, g >> 1) + 8 * (g & 1));
            }
          }
        };
        frag a[2][MBW];
        a_read(a[0], 0);
#pragma unroll
        for (int t = 0; t < T; ++t) {
          if (t + 1 < T) a_read(a[(t + 1) & 1], t + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mb = 0; mb < MBW; ++mb)
            if (rg * MBW + mb < nblk) {
#pragma unroll
              for (int i = 0; i < NBW; ++i)
                acc[mb][i] = FR::mma(bf[par][i][t], a[t & 1][mb], acc[mb][i]);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // (3) park the rows of phase kk+1 (issued one phase ago) in the other LDS buffer
      {
        char *sn = reinterpret_cast<char *>(s_a[par ^ 1]);
#pragma unroll
        for (int m = 0; m < GI; ++m) {
          const int it = tid + m * NT, p = it / CH, c = it % CH;
          if (ITEMS % NT == 0 || it < ITEMS)
            *reinterpret_cast<u32x4 *>(sn + p * ROWB + 16 * ATileB<CIN>::pos(p, c)) = greg[par ^ 1][m];
        }
      }
      __syncthreads();
    };
    // phases -2 and -1 only prime the pipeline (no rows to multiply yet)
    for (int kk = -2; kk < K; kk += 2) {
      phase(kk, std::integral_constant<int, 0>{});
      phase(kk + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue: folded BN, residual, ReLU, straight from the accumulators
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
      const int row = row0 + (rg * MBW + mb) * 16 + ln;
      if (row < row_end) {
#pragma unroll
        for (int i = 0; i < NBW; ++i)
          store_acc<COUT>(acc[mb][i], row, (cg * NBW + i) * 16 + 4 * g, scale, shift, residual, relu,
                          out, out_f32);
      }
    }
  }
}

// (K, cin, cout) f32 -> bf16 fragment order.  cinp >= 32: [k][nb][t][lane][8], element j =
// W[k][32t + 8(lane>>4) + j][16nb + (lane&15)]; cinp == 16: [k][nb][lane][4], element j =
// W[k][4(lane>>4) + j][16nb + (lane&15)].  Rows >= cin are zero.
__global__ void pack_filters_bf16_kernel(const float *__restrict__ w, int K, int cin, int cinp,
                                         int cout, __bf16 *__restrict__ wp) {
  const int E = cinp >= 32 ? 8 : 4, T = cinp >= 32 ? cinp / 32 : 1, NB = cout / 16;
  const long long total = (long long)K * NB * T * 64 * E;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(e % E);
    long long rest = e / E;
    const int lane = (int)(rest & 63);
    rest >>= 6;
    const int t = (int)(rest % T);
    rest /= T;
    const int nb = (int)(rest % NB);
    const int k = (int)(rest / NB);
    const int c = (E == 8 ? 32 * t + 8 * (lane >> 4) : 4 * (lane >> 4)) + j;
    const int n = 16 * nb + (lane & 15);
    wp[e] = (__bf16)((c < cin) ? w[((size_t)k * cin + c) * cout + n] : 0.0f);
  }
}

// src (n, c_src) f32 -> dst (n, c_dst) bf16 (round to nearest even), columns >= c_src zero
__global__ void cast_bf16_kernel(const float *__restrict__ src, int cs, __bf16 *__restrict__ dst,
                                 int cd, const int *n_dev, int n_bound) {
  const int n = load_n(n_dev, n_bound);
  const long long total = (long long)n * cd;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(e / cd), c = (int)(e - (long long)r * cd);
    dst[e] = (__bf16)(c < cs ? src[(size_t)r * cs + c] : 0.0f);
  }
}

struct ConvArgs {
  const void *feat;
  unsigned feat_bytes;
  const int *nbr;
  int nbr_stride, K, k_first;
  const void *wpk;
  const float *scale, *shift;
  const __bf16 *residual;
  int relu;
  void *out;
  int out_f32;
  const int *n_out_dev;
  int n_out_bound;
  hipStream_t st;
};

template <int CIN, int COUT, int MB, int D1>
int launch_direct(const ConvArgs &a) {
  sf_note_kernel("conv_bf16_direct_kernel<%d, %d, %d, %d>", CIN, COUT, MB, D1);
  hipLaunchKernelGGL((conv_bf16_direct_kernel<CIN, COUT, MB, D1>),
                     dim3((sf_div_up(a.n_out_bound, 16 * MB * 4) + 7) & ~7), dim3(256), 0, a.st, a.feat,
                     a.feat_bytes, a.nbr, a.nbr_stride, a.k_first, a.wpk, a.scale, a.shift,
                     a.residual, a.relu, a.out, a.out_f32, a.n_out_dev, a.n_out_bound);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

template <int CIN, int COUT, int TM, int CG>
int launch_tile(const ConvArgs &a) {
  constexpr int NB = COUT / 16;
  if constexpr (NB % CG != 0 || (TM / 16) % (4 / CG) != 0) {
    sf_set_error("bf16 tile shape tm=%d cg=%d does not fit cout=%d", TM, CG, COUT);
    return SF_ERR_BAD_ARG;
  } else {
    // plain tiling: one tile per workgroup, the grid a multiple of 8 (the balanced hand-out over
    // the resident workgroups measured no faster for these shapes)
    const int balance = 0;
    int grid = (sf_div_up(a.n_out_bound, TM) + 7) & ~7;
    if (grid > 8192) grid = 8192;
    sf_note_kernel("conv_bf16_tile_kernel<%d, %d, %d, %d, 4, 2>", CIN, COUT, TM, CG);
    hipLaunchKernelGGL((conv_bf16_tile_kernel<CIN, COUT, TM, CG, 4, 2>), dim3(grid), dim3(256), 0,
                       a.st, a.feat, a.feat_bytes, a.nbr, a.nbr_stride, a.K, a.k_first, a.wpk,
                       a.scale, a.shift, a.residual, a.relu, a.out, a.out_f32, a.n_out_dev,
                       a.n_out_bound, balance);
    SF_CHECK_LAUNCH();
    return SF_OK;
  }
}

// Measured choices (tools/sweep_bf16.sh at the time): the direct kernel (2 row blocks per wave,
// look-ahead 2) for the 27-offset layers up to 32 input channels, 64-row tiles with the widest
// column-group split elsewhere.
template <int CIN, int COUT>
int launch_cfg(const ConvArgs &a) {
  if constexpr (CIN <= 32 && COUT <= 64) {
    if (a.K == 27) return launch_direct<CIN, COUT, 2, 2>(a);
  }
  constexpr int NB = COUT / 16;
  constexpr int CGMAX = NB >= 4 ? 4 : NB;
  return launch_tile<CIN, COUT, 64, CGMAX>(a);
}

template <int CIN>
int dispatch_cout(int cout, const ConvArgs &a) {
  switch (cout) {
    case 16: return launch_cfg<CIN, 16>(a);
    case 32: return launch_cfg<CIN, 32>(a);
    case 64: return launch_cfg<CIN, 64>(a);
    case 128: return launch_cfg<CIN, 128>(a);
  }
  sf_set_error("unsupported cout=%d (16, 32, 64 or 128)", cout);
  return SF_ERR_BAD_ARG;
}

}  // namespace

extern "C" int sf_conv_padded_cin_bf16(int cin) {
  if (cin < 1) return -1;
  if (cin <= 16) return 16;
  if (cin <= 32) return 32;
  if (cin <= 64) return 64;
  if (cin <= 128) return 128;
  return -1;
}

extern "C" int sf_pack_filters_bf16(const float *w_kio, int K, int cin, int cout, void *w_packed,
                                    void *stream) {
  const int cp = sf_conv_padded_cin_bf16(cin);
  SF_CHECK_ARG(w_kio && w_packed, "filters pointer is NULL");
  SF_CHECK_ARG(K >= 1 && K <= kMaxK, "kernel volume %d not in [1,%d]", K, kMaxK);
  SF_CHECK_ARG(cp > 0, "unsupported cin=%d (1..128)", cin);
  SF_CHECK_ARG(cout == 16 || cout == 32 || cout == 64 || cout == 128, "unsupported cout=%d", cout);
  const long long total = (long long)K * cp * cout;
  hipLaunchKernelGGL(pack_filters_bf16_kernel, dim3(sf_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, w_kio, K, cin, cp, cout, (__bf16 *)w_packed);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_cast_f32_bf16(const float *src, int c_src, void *dst, int c_dst, const int *n_dev,
                                int n_bound, void *stream) {
  SF_CHECK_ARG(c_src >= 1 && c_dst >= c_src && n_bound >= 0, "bad channel counts / n_bound");
  if (n_bound == 0) return SF_OK;
  SF_CHECK_ARG(src && dst, "src / dst is NULL");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(sf_grid((long long)n_bound * c_dst, 256)), dim3(256), 0,
                     (hipStream_t)stream, src, c_src, (__bf16 *)dst, c_dst, n_dev, n_bound);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_sparse_conv_bf16(const void *features_in, int n_in_bound, int cin_padded,
                                   const int *nbr, int nbr_stride, int K, int k_first,
                                   const void *w_packed, int cout, const float *scale,
                                   const float *shift, const void *residual, int relu, void *out,
                                   int out_is_f32, const int *n_out_dev, int n_out_bound,
                                   void *stream) {
  SF_CHECK_ARG(K >= 1 && K <= kMaxK, "kernel volume %d not in [1,%d]", K, kMaxK);
  SF_CHECK_ARG(n_out_bound >= 0 && nbr_stride >= n_out_bound, "n_out_bound=%d nbr_stride=%d",
               n_out_bound, nbr_stride);
  SF_CHECK_ARG((scale == nullptr) == (shift == nullptr), "scale and shift go together");
  SF_CHECK_ARG(n_in_bound >= 0, "n_in_bound=%d", n_in_bound);
  if (n_out_bound == 0) return SF_OK;
  SF_CHECK_ARG(features_in && nbr && w_packed && out, "NULL tensor pointer");
  SF_CHECK_ARG(((uintptr_t)features_in & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
                   ((uintptr_t)w_packed & 15) == 0 && ((uintptr_t)residual & 7) == 0 &&
                   ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0,
               "feature / filter / epilogue pointers must be 16-byte aligned (residual: 8)");
  // the gather addresses rows with 32-bit byte offsets; an empty slot is the offset of row -1,
  // which must stay beyond the table
  const long long fb = (long long)n_in_bound * cin_padded * 2;
  SF_CHECK_ARG(fb < (1ll << 31), "input feature table of %lld bytes exceeds 2 GiB: split the batch",
               fb);
  ConvArgs a{features_in, (unsigned)fb, nbr, nbr_stride, K, k_first, w_packed, scale, shift,
             (const __bf16 *)residual, relu, out, out_is_f32, n_out_dev, n_out_bound,
             (hipStream_t)stream};
  switch (cin_padded) {
    case 16: return dispatch_cout<16>(cout, a);
    case 32: return dispatch_cout<32>(cout, a);
    case 64: return dispatch_cout<64>(cout, a);
    case 128: return dispatch_cout<128>(cout, a);
  }
  sf_set_error("unsupported cin_padded=%d (16, 32, 64 or 128; see sf_conv_padded_cin_bf16)",
               cin_padded);
  return SF_ERR_BAD_ARG;
}

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

// ---- row order: stable LSD radix sort of (27-bit mask, row), 3 passes of 9 bits ------------------
// Hand-written (no library call on the path): every kernel is bounded by the DEVICE-side row count
// (blocks past it exit at once), so a rulebook sized for its 8x geometric bound costs what its
// real rows cost.
constexpr int kSortChunk = 2048;   // keys per 256-thread block: 4 waves x 8 batches of 64
constexpr int kSortBins = 512;     // 9-bit digits
constexpr int kTileRows = 64;      // rows per tile of the ordered kernels (schedule granularity)

__global__ void row_mask_kernel(const int *__restrict__ nbr, int nbr_stride, int K,
                                const int *n_dev, int n_bound, unsigned *__restrict__ keys) {
  const int n = load_n(n_dev, n_bound);
  const int n_round = (n + kSortChunk - 1) / kSortChunk * kSortChunk;  // whole chunks are sorted
  const int end = n_round < n_bound ? n_round : n_bound;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < end; o += gridDim.x * blockDim.x) {
    unsigned m = 0xFFFFFFFFu;  // rows past the count sort behind every real row
    if (o < n) {
      m = 0;
      for (int k = 0; k < K; ++k) m |= (unsigned)(nbr[(size_t)k * nbr_stride + o] >= 0) << k;
    }
    keys[o] = m;
  }
}

// pass 1 of a digit: per-block histogram, hist[bin * nblk_bound + block]
__global__ void __launch_bounds__(256)
sort_hist_kernel(const unsigned *__restrict__ keys, const int *n_dev, int n_bound, int shift,
                 int nblk_bound, int *__restrict__ hist) {
  __shared__ int s_h[kSortBins];
  const int n = load_n(n_dev, n_bound);
  const int base = blockIdx.x * kSortChunk;
  if (base >= n) return;
  for (int i = threadIdx.x; i < kSortBins; i += 256) s_h[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < kSortChunk; i += 256) {
    const int o = base + i;
    const unsigned key = o < n_bound ? keys[o] : 0xFFFFFFFFu;
    atomicAdd(&s_h[(key >> shift) & (kSortBins - 1)], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kSortBins; i += 256) hist[(size_t)i * nblk_bound + blockIdx.x] = s_h[i];
}

// pass 2: exclusive scan of the (bin-major, block-minor) histogram over the ACTIVE blocks, one
// 1024-thread workgroup
__global__ void __launch_bounds__(1024)
sort_scan_kernel(const int *n_dev, int n_bound, int nblk_bound, int *__restrict__ hist) {
  __shared__ int smem[17];
  const int n = load_n(n_dev, n_bound);
  const int nblk = (n + kSortChunk - 1) / kSortChunk;
  const int total = nblk * kSortBins;
  const int per = (total + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = (lo + per < total) ? lo + per : total;
  int sum = 0;
  for (int e = lo; e < hi; ++e) sum += hist[(size_t)(e / nblk) * nblk_bound + e % nblk];
  int tot;
  int run = sf_block_excl_scan(sum, smem, &tot);
  for (int e = lo; e < hi; ++e) {
    int *p = &hist[(size_t)(e / nblk) * nblk_bound + e % nblk];
    const int v = *p;
    *p = run;
    run += v;
  }
}

// pass 3: stable scatter.  A wave owns 512 consecutive keys of the block's chunk and places them
// batch by batch (64 at a time, in order); lanes of one batch that share a digit are ranked by
// lane (a match over the 9 digit bits).
__global__ void __launch_bounds__(256)
sort_scatter_kernel(const unsigned *__restrict__ keys, const int *__restrict__ rows,
                    const int *n_dev, int n_bound, int shift, int nblk_bound,
                    const int *__restrict__ hist, unsigned *__restrict__ keys_out,
                    int *__restrict__ rows_out) {
  __shared__ int s_h[4][kSortBins];
  const int n = load_n(n_dev, n_bound);
  const int base = blockIdx.x * kSortChunk;
  if (base >= n) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * kSortBins; i += 256) (&s_h[0][0])[i] = 0;
  __syncthreads();
  unsigned key[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int o = base + wave * 512 + b * 64 + lane;
    key[b] = o < n_bound ? keys[o] : 0xFFFFFFFFu;
    atomicAdd(&s_h[wave][(key[b] >> shift) & (kSortBins - 1)], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kSortBins; i += 256) {
    int run = hist[(size_t)i * nblk_bound + blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int c = s_h[w][i];
      s_h[w][i] = run;
      run += c;
    }
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int o = base + wave * 512 + b * 64 + lane;
    const int d = (int)((key[b] >> shift) & (kSortBins - 1));
    unsigned long long m = ~0ull;
#pragma unroll
    for (int bit = 0; bit < 9; ++bit) {
      const unsigned long long bal = __ballot((d >> bit) & 1);
      m &= ((d >> bit) & 1) ? bal : ~bal;
    }
    const int rank = __popcll(m & lt);
    volatile int *cursor = &s_h[wave][d];
    const int pos = *cursor + rank;
    __builtin_amdgcn_wave_barrier();
    if (rank == 0) *cursor = pos + __popcll(m);
    __builtin_amdgcn_wave_barrier();
    if (o < n_bound && pos < n_bound) {
      keys_out[pos] = key[b];
      rows_out[pos] = rows ? rows[o] : o;
    }
  }
}

// Work of every tile of kTileRows sorted rows = number of (16-row block, offset) products it
// needs; one wave per tile.
__global__ void __launch_bounds__(256)
tile_weight_kernel(const unsigned *__restrict__ keys_sorted, const int *n_dev, int n_bound,
                   int *__restrict__ weight) {
  const int n = load_n(n_dev, n_bound);
  const int tiles = (n + kTileRows - 1) / kTileRows;
  const int lane = threadIdx.x & 63;
  for (int t = blockIdx.x * 4 + (threadIdx.x >> 6); t < tiles; t += gridDim.x * 4) {
    const int e = t * kTileRows + lane;
    unsigned m = e < n ? (keys_sorted[e] & 0x7FFFFFFu) : 0u;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) m |= (unsigned)__shfl_xor((int)m, d, 64);
    int w = (lane & 15) == 0 ? __popc(m) : 0;
    w += __shfl_xor(w, 16, 64);
    w += __shfl_xor(w, 32, 64);
    if (lane == 0) weight[t] = w;
  }
}

// Tile schedule = tiles by descending work (counting sort over the <= 109 distinct weights, one
// workgroup).  Dealt to the persistent workgroups in snake order it is a longest-processing-time
// assignment: the heaviest tiles are spread first, the light ones level the tails.
__global__ void __launch_bounds__(1024)
tile_schedule_kernel(const int *__restrict__ weight, const int *n_dev, int n_bound,
                     int *__restrict__ sched) {
  __shared__ int s_cnt[128];
  const int n = load_n(n_dev, n_bound);
  const int tiles = (n + kTileRows - 1) / kTileRows;
  if (threadIdx.x < 128) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < tiles; t += 1024) atomicAdd(&s_cnt[127 - weight[t]], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < 128; ++i) {
      const int c = s_cnt[i];
      s_cnt[i] = run;
      run += c;
    }
  }
  __syncthreads();
  // equal weights keep no particular order (the schedule affects timing only, never results)
  for (int t = threadIdx.x; t < tiles; t += 1024) sched[atomicAdd(&s_cnt[127 - weight[t]], 1)] = t;
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
  // tiles in the schedule's order (descending work), dealt to the persistent workgroups in snake
  // order: round j gives workgroup b entry j*G + b (even j) or j*G + G-1-b (odd j)
  const int *__restrict__ sched = order ? order + n_out_bound : nullptr;
  const int G = gridDim.x;

  for (int j = 0; j * G < tiles; ++j) {
    const int slot = j * G + ((j & 1) ? G - 1 - (int)blockIdx.x : (int)blockIdx.x);
    if (slot >= tiles) continue;
    const int tile = sched ? sched[slot] : slot;
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
  static_assert(TM == kTileRows, "the tile schedule is built for kTileRows-row tiles");
  static int slots = 0;          // resident workgroups on the device
  if (slots == 0) {
    int dev = 0, cus = 256, per_cu = 1;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, WAVES * 64, lds) != hipSuccess ||
        per_cu < 1)
      per_cu = 1;
    slots = cus * per_cu;
  }
  const int tiles = sf_div_up(a.n_out_bound, TM);
  const int grid = tiles < slots ? tiles : slots;
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

namespace {
size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
int sort_blocks(int n_bound) { return sf_div_up(n_bound > 0 ? n_bound : 1, kSortChunk); }
}  // namespace

extern "C" size_t sf_row_order_workspace(int n_bound) {
  if (n_bound < 1) n_bound = 1;
  const size_t arr = align_up((size_t)n_bound * 4);
  const size_t tiles = (size_t)sf_div_up(n_bound, kTileRows);
  // keys x2, rows x2, per-block histograms, tile weights
  return 4 * arr + align_up((size_t)sort_blocks(n_bound) * kSortBins * 4) + align_up(tiles * 4);
}

extern "C" int sf_row_order_size(int n_bound) {
  if (n_bound < 1) n_bound = 1;
  return n_bound + sf_div_up(n_bound, kTileRows);
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
  const size_t arr = align_up((size_t)n_bound * 4);
  const int nblk = sort_blocks(n_bound);
  char *ws = (char *)workspace;
  unsigned *keys0 = (unsigned *)ws, *keys1 = (unsigned *)(ws + arr);
  int *rows0 = (int *)(ws + 2 * arr), *rows1 = (int *)(ws + 3 * arr);
  int *hist = (int *)(ws + 4 * arr);
  int *weight = (int *)(ws + 4 * arr + align_up((size_t)nblk * kSortBins * 4));
  int *sched = row_order + n_bound;
  hipLaunchKernelGGL(row_mask_kernel, dim3(sf_grid(n_bound, 256)), dim3(256), 0, st, nbr, nbr_stride,
                     K, n_dev, n_bound, keys0);
  SF_CHECK_LAUNCH();
  // three stable passes over the 27 mask bits; the last one lands in row_order
  const unsigned *kin[3] = {keys0, keys1, keys0};
  unsigned *kout[3] = {keys1, keys0, keys1};
  const int *rin[3] = {nullptr, rows1, rows0};   // pass 0 starts from the identity
  int *rout[3] = {rows1, rows0, row_order};
  for (int p = 0; p < 3; ++p) {
    hipLaunchKernelGGL(sort_hist_kernel, dim3(nblk), dim3(256), 0, st, kin[p], n_dev, n_bound, 9 * p,
                       nblk, hist);
    SF_CHECK_LAUNCH();
    hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(1024), 0, st, n_dev, n_bound, nblk, hist);
    SF_CHECK_LAUNCH();
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblk), dim3(256), 0, st, kin[p], rin[p], n_dev,
                       n_bound, 9 * p, nblk, (const int *)hist, kout[p], rout[p]);
    SF_CHECK_LAUNCH();
  }
  const int tiles = sf_div_up(n_bound, kTileRows);
  hipLaunchKernelGGL(tile_weight_kernel, dim3(sf_grid(tiles, 4)), dim3(256), 0, st,
                     (const unsigned *)keys1, n_dev, n_bound, weight);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(tile_schedule_kernel, dim3(1), dim3(1024), 0, st, (const int *)weight, n_dev,
                     n_bound, sched);
  SF_CHECK_LAUNCH();
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

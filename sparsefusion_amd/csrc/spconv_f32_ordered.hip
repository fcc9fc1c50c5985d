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
#include <stdlib.h>

#include "front_internal.h"

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

// ---- the plan of a rulebook: row order + tile headers + sorted neighbour table --------------------
// Built once per rulebook (sf_row_order_by_mask), read by every layer that uses the rulebook:
//   plan[0, n_bound)                 row_order: output rows by ascending 27-bit neighbour mask (stable)
//   plan[n_bound, +T)                tile schedule: tiles by descending work (T = ceil(n_bound / 64))
//   + T x 32 ints                    tile headers: [0] number of phases np = max(active offsets, 3),
//                                    [1..np] the phases READY MADE in execution order (the centre
//                                    offset k_first first when active, then ascending):
//                                    k | 16-row block mask << 5 | flags << 9 (1 multiplies, 2 first,
//                                    4 last, 8 second phase of the tile, 16 valid), [28] work,
//                                    [29] 27-bit mask of the active offsets
//   + 27 x (64 T) ints               the neighbour table permuted into row_order (column per offset):
//                                    a tile reads 256 contiguous bytes per offset
// Hand-written (no library call on the path): every kernel is bounded by the DEVICE-side row count
// (blocks past it exit at once), so a rulebook sized for its 8x geometric bound costs what its
// real rows cost.  Stable LSD radix sort, 3 passes of 9 bits, four launches: masks + first
// histogram | scatter 1 (+ histogram 2) | scatter 2 (+ histogram 3) | scatter 3.  A scatter block
// derives its bin offsets from the per-chunk histogram table itself (a few hundred KB in L2), and
// counts the NEXT digit of every key into the chunk the key lands in, so no separate scan or
// histogram launch is needed.
constexpr int kSortChunk = 2048;   // keys per 256-thread block: 4 waves x 8 batches of 64
constexpr int kSortBins = 512;     // 9-bit digits
constexpr int kTileRows = 64;      // rows per tile of the ordered kernels
constexpr int kHdrInts = 32;
constexpr int kPlanMinPhases = 3;   // tiles are padded to this many phases (the kernels look that far ahead)

// zeroes the histogram rows of the chunks in use (the tables are sized for the bound)
__global__ void __launch_bounds__(256)
hist_zero_kernel(const int *n_dev, int n_bound, int *__restrict__ hist, size_t table_ints) {
  const int n = load_n(n_dev, n_bound);
  const int used = (n + kSortChunk - 1) / kSortChunk * kSortBins;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < used; i += gridDim.x * 256) {
    hist[i] = 0;
    hist[table_ints + i] = 0;
    hist[2 * table_ints + i] = 0;
  }
}

__global__ void __launch_bounds__(256)
mask_hist_kernel(const int *__restrict__ nbr, int nbr_stride, int K, const int *n_dev, int n_bound,
                 unsigned *__restrict__ keys, int *__restrict__ hist) {
  __shared__ int s_h[kSortBins];
  const int n = load_n(n_dev, n_bound);
  const int base = blockIdx.x * 256;             // eight blocks share a sort chunk
  if (base >= (n + kSortChunk - 1) / kSortChunk * kSortChunk) return;
  for (int i = threadIdx.x; i < kSortBins; i += 256) s_h[i] = 0;
  __syncthreads();
  const int o = base + threadIdx.x;
  unsigned m = 0xFFFFFFFFu;  // rows past the count sort behind every real row
  if (o < n) {
    int v[kMaxK];
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) v[k] = k < K ? nbr[(size_t)k * nbr_stride + o] : -1;
    m = 0;
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) m |= (unsigned)(v[k] >= 0) << k;
  }
  if (o < n_bound) keys[o] = m;
  atomicAdd(&s_h[m & (kSortBins - 1)], 1);
  __syncthreads();
  int *h = hist + (size_t)(base / kSortChunk) * kSortBins;
  for (int i = threadIdx.x; i < kSortBins; i += 256)
    if (s_h[i]) atomicAdd(&h[i], s_h[i]);
}

// One stable pass.  hist[chunk][bin] = keys of this pass's digit `bin` in `chunk`.  A wave owns 512
// consecutive keys of the block's chunk and places them batch by batch (64 at a time, in order);
// lanes of one batch that share a digit are ranked by lane (a match over the 9 digit bits).
__global__ void __launch_bounds__(256)
sort_scatter_kernel(const unsigned *__restrict__ keys, const int *__restrict__ rows,
                    const int *n_dev, int n_bound, int shift, const int *__restrict__ hist,
                    unsigned *__restrict__ keys_out, int *__restrict__ rows_out,
                    int *__restrict__ hist_next) {
  __shared__ int s_h[4][kSortBins];
  __shared__ int s_base[kSortBins];
  __shared__ int smem[17];
  const int n = load_n(n_dev, n_bound);
  const int base = blockIdx.x * kSortChunk;
  if (base >= n) return;
  const int nchunk = (n + kSortChunk - 1) / kSortChunk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * kSortBins; i += 256) (&s_h[0][0])[i] = 0;
  // keys first (their latency hides behind the histogram sums)
  unsigned key[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int o = base + wave * 512 + b * 64 + lane;
    key[b] = o < n_bound ? keys[o] : 0xFFFFFFFFu;
  }
  // bins 2t and 2t+1: keys of the bin in all chunks, and in the chunks before this one
  int tot[2] = {0, 0}, pre[2] = {0, 0};
  {
    const int2 *h2 = reinterpret_cast<const int2 *>(hist) + threadIdx.x;
    for (int c0 = 0; c0 < nchunk; c0 += 16) {     // 16 independent loads per round trip
      int2 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int c = c0 + u < nchunk ? c0 + u : nchunk - 1;
        v[u] = h2[(size_t)c * (kSortBins / 2)];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int live = c0 + u < nchunk ? 1 : 0, before = c0 + u < (int)blockIdx.x ? 1 : 0;
        tot[0] += live * v[u].x;
        tot[1] += live * v[u].y;
        pre[0] += before * v[u].x;
        pre[1] += before * v[u].y;
      }
    }
  }
  int all;
  const int excl = sf_block_excl_scan(tot[0] + tot[1], smem, &all);  // (has the barrier s_h needs)
  s_base[2 * threadIdx.x] = excl + pre[0];
  s_base[2 * threadIdx.x + 1] = excl + tot[0] + pre[1];
#pragma unroll
  for (int b = 0; b < 8; ++b) atomicAdd(&s_h[wave][(key[b] >> shift) & (kSortBins - 1)], 1);
  __syncthreads();
  for (int i = threadIdx.x; i < kSortBins; i += 256) {
    int run = s_base[i];
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
      if (keys_out) keys_out[pos] = key[b];
      rows_out[pos] = rows ? rows[o] : o;
    }
    if (hist_next) {
      // next digit of the key, counted into the chunk the key lands in.  Lanes of the batch that
      // hit the same counter (the rule: equal keys land side by side) add once, through a leader;
      // one atomic per key made the popular masks serialise at the L2.
      const bool ok = o < n_bound && pos < n_bound;
      const unsigned cell = ok ? (unsigned)(pos / kSortChunk) * kSortBins +
                                     ((key[b] >> (shift + 9)) & (kSortBins - 1))
                               : 0xFFFFFFFFu;
      unsigned long long mm = ~0ull;
#pragma unroll
      for (int bit = 0; bit < 32; ++bit) {
        const unsigned long long bal = __ballot((cell >> bit) & 1u);
        mm &= ((cell >> bit) & 1u) ? bal : ~bal;
      }
      if (ok && (mm & lt) == 0) atomicAdd(&hist_next[cell], (int)__popcll(mm));
    }
  }
}

// One wave per tile of 64 sorted rows: permuted neighbour table, active offsets with their 16-row
// block masks, work of the tile (number of (offset, block) products).
__global__ void __launch_bounds__(256)
tile_plan_kernel(const int *__restrict__ nbr, int nbr_stride, int K, int k_first, const int *n_dev,
                 int n_bound, int *__restrict__ order, int *__restrict__ hdr, int *__restrict__ nbrs,
                 int s_stride) {
  const int n = load_n(n_dev, n_bound);
  const int tiles = (n + kTileRows - 1) / kTileRows;
  const int lane = threadIdx.x & 63;
  for (int t = blockIdx.x * 4 + (threadIdx.x >> 6); t < tiles; t += gridDim.x * 4) {
    const int e = t * kTileRows + lane;
    const int row = e < n ? order[e] : -1;
    if (e >= n && e < n_bound) order[e] = -1;
    int mybm = 0;
    int vv[kMaxK];
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) vv[k] = (k < K && row >= 0) ? nbr[(size_t)k * nbr_stride + row] : -1;
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
      const int v = vv[k];
      if (k < K) nbrs[(size_t)k * s_stride + e] = v;
      const unsigned long long bal = __ballot(v >= 0);
      int bm = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) bm |= (int)(((bal >> (16 * b)) & 0xFFFFull) != 0) << b;
      if (lane == k) mybm = bm;
    }
    // lane k < 27 holds the block mask of offset k.  Phases in execution order: the centre offset
    // first (when it is active), then the active offsets ascending, then padding up to
    // kPlanMinPhases (the reference's summation order, spconv_ops.h:300-356).
    const unsigned act = (unsigned)__ballot(mybm != 0);
    const int nact = __popc(act);
    const int np = nact > kPlanMinPhases ? nact : kPlanMinPhases;
    const bool centre = k_first >= 0 && k_first < K && ((act >> k_first) & 1u);
    const int pc = centre ? __popc(act & ((1u << k_first) - 1u)) : -1;   // ascending rank of the centre
    int *h = hdr + (size_t)t * kHdrInts;
    int pos = -1, word = 0;
    if (lane < 32 && mybm != 0) {                 // an active offset: its phase
      const int rank = __popc(act & ((1u << lane) - 1u));
      pos = !centre ? rank : (lane == k_first ? 0 : (rank < pc ? rank + 1 : rank));
      word = lane | (mybm << 5) | (1 << 9);
    } else if (lane >= 32 && lane - 32 >= nact && lane - 32 < np) {   // a padding phase
      pos = lane - 32;
    }
    if (pos >= 0)
      h[1 + pos] = word | (((pos == 0 ? 2 : 0) | (pos == np - 1 ? 4 : 0) | (pos == 1 ? 8 : 0) | 16) << 9);
    int w = __popc(mybm);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) w += __shfl_xor(w, d, 64);
    if (lane == 0) {
      h[0] = np;
      // cost of the tile for the schedule: a phase costs about three block products of fixed work
      // (descriptor, loads, barrier) besides its products; balancing products alone left the
      // workgroups up to 1.5x apart in phases (tools: LPT simulation on the frame's real masks)
      h[28] = w + 3 * nact;
      h[29] = (int)act;
    }
  }
}

// Tile schedule = tiles by descending cost (counting sort over the <= 190 distinct values, one
// workgroup).  Dealt to the persistent workgroups in snake order it is a longest-processing-time
// assignment: the heaviest tiles are spread first, the light ones level the tails.
__global__ void __launch_bounds__(1024)
tile_schedule_kernel(const int *__restrict__ hdr, const int *n_dev, int n_bound,
                     int *__restrict__ sched) {
  __shared__ int s_cnt[256];   // cost <= 108 + 3 * 27
  const int n = load_n(n_dev, n_bound);
  const int tiles = (n + kTileRows - 1) / kTileRows;
  if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < tiles; t += 1024) atomicAdd(&s_cnt[255 - hdr[(size_t)t * kHdrInts + 28]], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) {
      const int c = s_cnt[i];
      s_cnt[i] = run;
      run += c;
    }
  }
  __syncthreads();
  // equal weights keep no particular order (the schedule affects timing only, never results)
  for (int t = threadIdx.x; t < tiles; t += 1024)
    sched[atomicAdd(&s_cnt[255 - hdr[(size_t)t * kHdrInts + 28]], 1)] = t;
}

struct PlanView {
  const int *order, *sched, *hdr, *nbrs;
  int s_stride;
};

__host__ __device__ inline int plan_tiles(int n_bound) { return (n_bound + kTileRows - 1) / kTileRows; }

// ---- the convolution ---------------------------------------------------------------------------
// A persistent workgroup walks its tiles (snake deal of the schedule) as ONE stream of phases --
// a phase = one active offset of one tile -- so that the loads of the next tile's first offsets
// are in flight while the current tile still multiplies: per phase q the workgroup
//   (1) issues the gathers of phase q+2 (raw buffer loads; a row without partner is an out-of-range
//       offset: zeros, no memory access, no branch), whose row indices it loaded one phase earlier,
//   (2) loads the row indices of phase q+3 from the sorted neighbour table (256 contiguous bytes),
//   (3) loads the filter fragments of phase q+1,
//   (4) multiplies phase q from LDS (the 16-row blocks that have a partner under this offset),
//   (5) parks the rows of phase q+1 in the other LDS buffer;  one barrier.
// Tiles are padded to three phases so that the look-ahead crosses at most one tile boundary; tile
// headers travel through a ring of three LDS slots, loaded two tiles ahead.
template <int CIN, int COUT, int CG, int WAVES, int MINB, int BFD>
__global__ void __launch_bounds__(WAVES * 64, MINB)
conv_f32_ordered_kernel(const float *__restrict__ feat, unsigned feat_bytes, PlanView pv, int K,
                        int k_first, const f32x4 *__restrict__ wp, const float *__restrict__ scale,
                        const float *__restrict__ shift, const float *__restrict__ residual, int relu,
                        float *__restrict__ out, const int *n_out_dev, int n_out_bound,
                        unsigned long long *__restrict__ prof) {
  constexpr int TM = kTileRows, NT = WAVES * 64, T = CIN / 16, ROWB = CIN * 4, CH = CIN / 4;
  constexpr int NB = COUT / 16, NBW = NB / CG, RG = WAVES / CG, MBT = TM / 16, MBW = MBT / RG;
  constexpr int ITEMS = TM * CH, GI = ITEMS / NT;
  constexpr int kRun = 128;   // tiles per pipeline run
  // (look-ahead in phases = kPlanMinPhases, the least phases per tile: the plan pads shorter tiles)
  static_assert(NB % CG == 0 && WAVES % CG == 0 && MBT % RG == 0 && CIN % 16 == 0 && ITEMS % NT == 0,
                "tile shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int BUFB = TM * ROWB;                                             // bytes of one row buffer
  int *s_hdr = reinterpret_cast<int *>(lds_raw + (size_t)2 * BUFB);           // [3][32]
  int *s_tiles = s_hdr + 3 * kHdrInts;                                        // [kRun]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % CG, rg = wave / CG;
  const int g = lane >> 4, ln = lane & 15;
  const int n_out = load_n(n_out_dev, n_out_bound);
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(feat), 0, feat_bytes, 0x00020000);
  (void)k_first;   // the summation order (centre first) is part of the plan: sf_row_order_by_mask
  const int tiles = (n_out + TM - 1) / TM;
  const int G = gridDim.x, b = blockIdx.x;
  // snake deal: round i gives this workgroup schedule entry i*G + b (even i) or i*G + G-1-b (odd i)
  const int full = tiles / G, rem = tiles - full * G;
  const int my_n = full + ((((full & 1) ? G - 1 - b : b) < rem) ? 1 : 0);

  f32x4 acc[MBW][NBW];
  u32x4 greg[2][GI];
  f32x4 bf[BFD][NBW][T];   // BFD = 1: one filter set, reloaded behind the multiply that used it
  int ireg[GI], orow[MBW];
#pragma unroll
  for (int m = 0; m < GI; ++m) {
    greg[0][m] = greg[1][m] = (u32x4){0u, 0u, 0u, 0u};
    ireg[m] = -1;
  }
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) {
    orow[mb] = -1;
#pragma unroll
    for (int i = 0; i < NBW; ++i) acc[mb][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  unsigned long long pt[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long tk0 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long tr0 = prof ? __builtin_amdgcn_s_memrealtime() : 0ull;
  for (int run0 = 0; run0 < my_n; run0 += kRun) {
    const int cnt = (my_n - run0 < kRun) ? my_n - run0 : kRun;
    __syncthreads();   // the previous run is drained
    if (tid < cnt) {
      const int i = run0 + tid;
      s_tiles[tid] = pv.sched[(size_t)i * G + ((i & 1) ? G - 1 - b : b)];
    }
    __syncthreads();
    if (tid < 2 * kHdrInts && (tid >> 5) < cnt)
      s_hdr[tid] = pv.hdr[(size_t)s_tiles[tid >> 5] * kHdrInts + (tid & 31)];
    __syncthreads();

    // phase descriptors: tile, info = k | block mask << 5 | flags << 9
    //   flags: 1 multiply, 2 first phase of its tile, 4 last phase of its tile, 8 second phase
    // The generator's tile header sits in a register (lane i holds entry i), so a descriptor costs a
    // few scalar instructions and two v_readlane, no LDS round trip on the phase's critical path.
    int gj = 0, ga = 0;        // generator: tile (within the run) and phase of the NEXT descriptor
    int hcur = s_hdr[lane & 31], g_tile = __builtin_amdgcn_readfirstlane(s_tiles[0]);
    auto next_desc = [&](int &d_tile, int &d_info) {
      d_tile = 0;
      d_info = 0;
      if (gj < cnt) {
        // the plan holds the tile's phases ready made, in execution order (tile_plan_kernel)
        const int np = __builtin_amdgcn_readlane(hcur, 0);
        d_tile = g_tile;
        d_info = __builtin_amdgcn_readlane(hcur, 1 + ga);
        if (++ga == np) {
          ++gj;
          ga = 0;
          if (gj < cnt) {
            hcur = s_hdr[(gj % 3) * kHdrInts + (lane & 31)];
            g_tile = __builtin_amdgcn_readfirstlane(s_tiles[gj]);
          }
        }
      }
    };
    int tA = 0, iA = 0, tB = 0, iB = 0, tC = 0, iC = 0, tD = 0, iD = 0;
    int cj = -1;               // tile (within the run) of the current phase
    int hreg = 0;

    auto phase = [&](auto PAR) {
      constexpr int par = decltype(PAR)::value;
      const unsigned long long ts0 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
      tA = tB; iA = iB; tB = tC; iB = iC; tC = tD; iC = iD;
      next_desc(tD, iD);
      // load order matters (the counter of outstanding loads retires in order): row indices of
      // phase q+3 first, then the filters of phase q+1, then the gathers of phase q+2 -- so the
      // wait for the filters (next phase's MFMAs) and for the indices (next phase's gathers) never
      // retires the gathers early; they are only waited for where they are parked, at the end of
      // the NEXT phase (a gather that misses the L2 takes longer than one phase).
      {
        int inew[GI];
        const int *ip = pv.nbrs + (size_t)(iD & 31) * pv.s_stride + (size_t)tD * TM;
#pragma unroll
        for (int m = 0; m < GI; ++m) inew[m] = ip[(tid + m * NT) / CH];
        if constexpr (BFD == 2) {
          const int k1 = iB & 31;
#pragma unroll
          for (int i = 0; i < NBW; ++i)
#pragma unroll
            for (int t = 0; t < T; ++t)
              bf[par ^ 1][i][t] = wp[(((size_t)k1 * NB + (cg * NBW + i)) * T + t) * 64 + lane];
        }
        const bool live = (iC >> 9) & 1;
#pragma unroll
        for (int m = 0; m < GI; ++m) {
          const int c = (tid + m * NT) % CH;
          const int src = live ? ireg[m] : -1;   // -1: out of range, zeros, no fetch
          greg[par][m] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)src * ROWB + c * 16, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < GI; ++m) ireg[m] = inew[m];
      }
      const int fl = iA >> 9;
      const unsigned long long ts1 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
      if (fl & 2) {            // first phase of a tile: fresh sums, its output rows, header of tile +2
        ++cj;
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb) {
          const int e = tA * TM + (rg * MBW + mb) * 16 + ln;
          orow[mb] = e < n_out ? pv.order[e] : -1;
#pragma unroll
          for (int i = 0; i < NBW; ++i) acc[mb][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (tid < kHdrInts && cj + 2 < cnt) hreg = pv.hdr[(size_t)s_tiles[cj + 2] * kHdrInts + tid];
      }
      // (3) this phase: the row blocks that have a partner under this offset.  In mask order
      //     nearly every phase has all of the wave's blocks active: that case is a straight line
      //     (a branch per block and MFMA measured a wait in front of every MFMA)
      if (fl & 1) {
        const int bmw = (iA >> (5 + rg * MBW)) & ((1 << MBW) - 1);
        const char *sa = reinterpret_cast<const char *>(lds_raw) + par * BUFB;
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
                  acc[mb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[BFD == 2 ? par : 0][i][t][j], a[t & 1][mb][j],
                                                                   acc[mb][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          if (bmw != 0) {
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
                    acc[mb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[BFD == 2 ? par : 0][i][t][j], av[j],
                                                                     acc[mb][i], 0, 0, 0);
              }
            }
          }
        }
      }
      if constexpr (BFD == 1) {   // the one filter set is free again: filters of phase q+1
        const int k1 = iB & 31;
#pragma unroll
        for (int i = 0; i < NBW; ++i)
#pragma unroll
          for (int t = 0; t < T; ++t)
            bf[0][i][t] = wp[(((size_t)k1 * NB + (cg * NBW + i)) * T + t) * 64 + lane];
      }
      const unsigned long long ts2 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
      if (fl & 4) {
        // ---- end of a tile: folded BN, residual, ReLU, straight from the accumulators; every
        //      load first, so the tile end costs one round trip ----------------------------------
        f32x4 sc[NBW], sh[NBW], rsd[MBW][NBW];
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
          const int c = (cg * NBW + i) * 16 + 4 * g;
          sc[i] = scale ? *reinterpret_cast<const f32x4 *>(scale + c) : (f32x4){1.f, 1.f, 1.f, 1.f};
          sh[i] = scale ? *reinterpret_cast<const f32x4 *>(shift + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int mb = 0; mb < MBW; ++mb) {
            const int row = orow[mb] < 0 ? 0 : orow[mb];
            rsd[mb][i] = residual ? *reinterpret_cast<const f32x4 *>(residual + (size_t)row * COUT + c)
                                  : (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb) {
          const int row = orow[mb];
#pragma unroll
          for (int i = 0; i < NBW; ++i) {
            const int c = (cg * NBW + i) * 16 + 4 * g;
            f32x4 v = acc[mb][i];
            if (scale) v = v * sc[i] + sh[i];
            if (residual) v += rsd[mb][i];
            if (relu) {
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
            }
            if (row >= 0) *reinterpret_cast<f32x4 *>(out + (size_t)row * COUT + c) = v;
          }
        }
      }
      if ((fl & 8) && tid < kHdrInts && cj + 2 < cnt) s_hdr[((cj + 2) % 3) * kHdrInts + tid] = hreg;
      // park the rows of phase q+1 (issued one phase ago) in the other LDS buffer
      {
        char *sn = reinterpret_cast<char *>(lds_raw) + (par ^ 1) * BUFB;
#pragma unroll
        for (int m = 0; m < GI; ++m) {
          const int it = tid + m * NT, p = it / CH, c = it % CH;
          *reinterpret_cast<u32x4 *>(sn + p * ROWB + 16 * ATileF<CIN>::pos(p, c)) = greg[par ^ 1][m];
        }
      }
      const unsigned long long ts3 = prof ? __builtin_amdgcn_s_memtime() : 0ull;
      __syncthreads();
      if (prof) {
        const unsigned long long ts4 = __builtin_amdgcn_s_memtime();
        pt[0] += ts1 - ts0;   // park + load issue
        pt[1] += ts2 - ts1;   // first-phase work + MFMA section
        pt[2] += ts3 - ts2;   // epilogue
        pt[3] += ts4 - ts3;   // barrier
        pt[4] += 1;           // phases
        pt[5] += (fl & 1);    // multiplying phases
      }
    };
    do {
      phase(std::integral_constant<int, 0>{});
      phase(std::integral_constant<int, 1>{});
    } while (((iA | iB | iC | iD) >> 13) & 1);
  }
  if (prof && (tid & 63) == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(&prof[i], pt[i]);
    atomicAdd(&prof[6], (unsigned long long)__builtin_amdgcn_s_memtime() - tk0);
    atomicAdd(&prof[7], 1ull);
    atomicAdd(&prof[8], (unsigned long long)__builtin_amdgcn_s_memrealtime() - tr0);   // 100 MHz ticks
  }
}

// One 16-byte-per-lane gather straight into LDS: lane L's chunk lands at lds + 16 L.  (The host pass
// of hipcc cannot type the LDS address-space cast and would silently drop the kernel's stub, so the
// body exists in the device pass only.)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, unsigned char *lds, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds, 16, voff,
                                           0, 0, 0);
#else
  (void)rs; (void)lds; (void)voff;
#endif
}

// ---- the same kernel with the gathers written straight into LDS (LDS-DMA) ----------------------
// `buffer_load_dwordx4 ... lds`: a wave-instruction writes 1 KiB of LDS linearly (lane L -> base +
// 16 L), the per-lane SOURCE offset makes it a row gather, and an out-of-range source still writes
// zeros.  No staging registers: the rows of phase q+2 go from L2 into one of THREE LDS buffers while
// phases q and q+1 multiply from the other two, and the kernel fits 3 workgroups per CU (the
// register-staged version needs 178 VGPRs at 64 channels).  The XOR swizzle of the LDS tile moves to
// the source side: lane (row p, slot s) fetches chunk s ^ swz(p), the MFMA side reads chunk c at slot
// c ^ swz(p) as before.  The three buffers are three __shared__ objects addressed with compile-time
// indices (the phase loop is unrolled by three), so the compiler's alias analysis keeps the reads of
// one buffer from waiting for the DMA into another; the only wait for the DMA is the hand-counted
// s_waitcnt vmcnt(N) in front of the raw s_barrier that ends a phase (a __syncthreads() there
// would drain every DMA in flight): N = the memory operations a wave issues after the DMA of phase
// q+1 (issued one phase ago).  Conditional extras (tile start / end) only make the wait stronger.
template <int CIN, int COUT, int CG, int WAVES, int MINB>
__global__ void __launch_bounds__(WAVES * 64, MINB)
conv_f32_dma_kernel(const float *__restrict__ feat, unsigned feat_bytes, PlanView pv, int K,
                    int k_first, const f32x4 *__restrict__ wp, const float *__restrict__ scale,
                    const float *__restrict__ shift, const float *__restrict__ residual, int relu,
                    float *__restrict__ out, const int *n_out_dev, int n_out_bound,
                    unsigned long long *__restrict__ wg_times) {
  constexpr int TM = kTileRows, T = CIN / 16, ROWB = CIN * 4, CH = CIN / 4;
  constexpr int NB = COUT / 16, NBW = NB / CG, RG = WAVES / CG, MBT = TM / 16, MBW = MBT / RG;
  constexpr int RPI = 1024 / ROWB;             // rows one DMA instruction covers
  constexpr int IPW = TM / RPI / WAVES;        // DMA instructions per wave and phase
  constexpr int NBL = NBW * T;                 // filter loads per wave and phase
  constexpr int kRun = 128;
  // memory operations a wave issues AFTER the DMA of phase q+1 (issued in phase q-1) until the end
  // of phase q: last phase's filters (loaded behind its multiply), this phase's indices, DMA, filters
  constexpr int kWaitN = 2 * IPW + 2 * NBL;
  static_assert(NB % CG == 0 && WAVES % CG == 0 && MBT % RG == 0 && CIN >= 64 && (TM / RPI) % WAVES == 0 &&
                    kWaitN < 64, "tile shape");

  __shared__ __attribute__((aligned(1024))) unsigned char s_b0[TM * ROWB];
  __shared__ __attribute__((aligned(1024))) unsigned char s_b1[TM * ROWB];
  __shared__ __attribute__((aligned(1024))) unsigned char s_b2[TM * ROWB];
  __shared__ int s_hdr[3 * kHdrInts];
  __shared__ int s_tiles[kRun];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % CG, rg = wave / CG;
  const int g = lane >> 4, ln = lane & 15;
  const int n_out = load_n(n_out_dev, n_out_bound);
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(feat), 0, feat_bytes, 0x00020000);
  (void)k_first;   // the summation order (centre first) is part of the plan: sf_row_order_by_mask
  const int tiles = (n_out + TM - 1) / TM;
  const int G = gridDim.x, b = blockIdx.x;
  // The schedule (tiles by descending cost) is dealt in rounds, snake order: `full` rounds of whole
  // tiles, then ONE more round for the `rem` tiles left over (the lightest ones): as many of them as
  // it takes are dealt as HALF tiles (16-row blocks 0-1 / 2-3 of the tile) so that the round has an
  // entry for every workgroup, or, when rem <= G / 2, all of them are.  With 1.5 tiles per workgroup
  // (frames 1-3 of the synthetic stream at stage 4: 380 tiles on 256 workgroups) half of the
  // workgroups held two tiles and the others one: mean busy 79-83 % of the launch.  A half tile pays
  // the per-phase work of a whole one for half the products, so no more tiles are split than needed.
  // Only with ONE workgroup per CU: with three (64 channels) an early finisher speeds up its CU mates
  // and the halves' extra phases cost more than they level (measured 117.0 vs 114.9 us).
  constexpr bool kSplit = MINB == 1;
  const int full = tiles / G, rem = tiles - full * G;
  const int whole = !kSplit ? rem : (2 * rem > G ? 2 * rem - G : 0);   // leftover tiles dealt whole (the heavier ones)
  const int last_entries = whole + 2 * (rem - whole);     // = G when rem > G / 2, else 2 * rem
  const int snake_last = (full & 1) ? G - 1 - b : b;
  const int my_n = full + (snake_last < last_entries ? 1 : 0);
  // entry i of this workgroup: tile | owned 16-row blocks << 24
  auto entry = [&](int i) -> int {
    if (i < full) return pv.sched[(size_t)i * G + ((i & 1) ? G - 1 - b : b)] | (0xF << 24);
    if (snake_last < whole) return pv.sched[(size_t)full * G + snake_last] | (0xF << 24);
    const int h = snake_last - whole;
    return pv.sched[(size_t)full * G + whole + (h >> 1)] | (((h & 1) ? 0xC : 0x3) << 24);
  };
  if (wg_times && tid == 0) wg_times[b] = wall_clock64();

  f32x4 acc[MBW][NBW];
  f32x4 bf[NBW][T];
  int ireg[IPW], orow[MBW];
  unsigned soff[IPW];        // source offset of this lane's chunk inside its row, per instruction
  int prow[IPW];             // tile row this lane fetches, per instruction
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int p = (wave * IPW + j) * RPI + lane / CH, slot = lane % CH;
    prow[j] = p;
    soff[j] = 16u * (unsigned)ATileF<CIN>::pos(p, slot);
    ireg[j] = -1;
  }
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) {
    orow[mb] = -1;
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
      acc[mb][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
#pragma unroll
  for (int i = 0; i < NBW; ++i)
#pragma unroll
    for (int t = 0; t < T; ++t) bf[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int run0 = 0; run0 < my_n; run0 += kRun) {
    const int cnt = (my_n - run0 < kRun) ? my_n - run0 : kRun;
    // the previous run is drained: its last (padding) phases still have zero-fill DMAs in flight,
    // and nothing the compiler knows of makes the barrier below wait for them
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
    __syncthreads();
    if (tid < cnt) s_tiles[tid] = entry(run0 + tid);
    // (every barrier of this kernel follows an explicit vector-memory wait: the set-up barriers a
    // full drain, the phase barriers the counted one -- tests/test_isa_dma_wait.py holds the ISA to it)
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    if (tid < 2 * kHdrInts && (tid >> 5) < cnt)
      s_hdr[tid] = pv.hdr[(size_t)(s_tiles[tid >> 5] & 0xFFFFFF) * kHdrInts + (tid & 31)];
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();

    int gj = 0, ga = 0;
    int hcur = s_hdr[lane & 31], g_tile = __builtin_amdgcn_readfirstlane(s_tiles[0]);
    auto next_desc = [&](int &d_tile, int &d_info) {
      d_tile = 0;
      d_info = 0;
      if (gj < cnt) {
        // the plan holds the tile's phases ready made, in execution order (tile_plan_kernel)
        const int np = __builtin_amdgcn_readlane(hcur, 0);
        d_tile = g_tile;
        d_info = __builtin_amdgcn_readlane(hcur, 1 + ga);
        if (++ga == np) {
          ++gj;
          ga = 0;
          if (gj < cnt) {
            hcur = s_hdr[(gj % 3) * kHdrInts + (lane & 31)];
            g_tile = __builtin_amdgcn_readfirstlane(s_tiles[gj]);
          }
        }
      }
    };
    int tA = 0, iA = 0, tB = 0, iB = 0, tC = 0, iC = 0, tD = 0, iD = 0;
    int cj = -1, hreg = 0;

    auto phase = [&](auto PH) {
      constexpr int ph = decltype(PH)::value;              // q mod 3: buffer of this phase's rows
      unsigned char *const cur = ph == 0 ? s_b0 : (ph == 1 ? s_b1 : s_b2);
      unsigned char *const nxt2 = ph == 0 ? s_b2 : (ph == 1 ? s_b0 : s_b1);   // buffer of phase q+2
      tA = tB; iA = iB; tB = tC; iB = iC; tC = tD; iC = iD;
      next_desc(tD, iD);
      // The phase's own loads -- indices of phase q+3, then the rows of phase q+2 -> LDS (vmcnt
      // retires in order: indices first, so that next phase's wait for them does not retire the DMA
      // early) -- are issued BEHIND the first group of MFMAs of a full phase: right after the
      // barrier both waves of a SIMD would spend ~400 cycles on them with the matrix pipe idle
      // (measured per wave with s_memtime stamps), one group later each does it while the other
      // multiplies.
      auto issue = [&]() {
        int inew[IPW];
        const int *ip = pv.nbrs + (size_t)(iD & 31) * pv.s_stride + (size_t)(tD & 0xFFFFFF) * TM;
#pragma unroll
        for (int j = 0; j < IPW; ++j) inew[j] = ip[prow[j]];
        // a padding phase fetches row -1: out of range, zeros land in LDS, no fetch.  (An OR with a
        // scalar mask, not a select: the select became a branch around the first DMA.)  So do the
        // rows of 16-row blocks a half tile does not own.
        const unsigned dead = ((iC >> 9) & 1) ? 0u : 0xFFFFFFFFu;
        const unsigned own = (unsigned)tC >> 24;
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
          const unsigned skip = dead | (((own >> (prow[j] >> 4)) & 1u) - 1u);
          lds_dma16(rs, nxt2 + (wave * IPW + j) * 1024, ((unsigned)ireg[j] | skip) * ROWB + soff[j]);
        }
        // compiler fence (no instruction): the memory operations below stay below the DMA, the
        // order kWaitN counts on (tests/test_isa_dma_wait.py checks the emitted ISA)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < IPW; ++j) ireg[j] = inew[j];
      };
      const int fl = iA >> 9;
      if (fl & 2) {            // first phase of a tile
        ++cj;
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb) {
          const int e = (tA & 0xFFFFFF) * TM + (rg * MBW + mb) * 16 + ln;
          orow[mb] = (e < n_out && (((unsigned)tA >> (24 + rg * MBW + mb)) & 1u)) ? pv.order[e] : -1;
#pragma unroll
          for (int i = 0; i < NBW; ++i) acc[mb][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (tid < kHdrInts && cj + 2 < cnt)
          hreg = pv.hdr[(size_t)(s_tiles[cj + 2] & 0xFFFFFF) * kHdrInts + tid];
      }
      const int bmw = (fl & 1) ? ((iA >> 5) & ((unsigned)tA >> 24) & 15) >> (rg * MBW) & ((1 << MBW) - 1) : 0;
      {
        auto a_at = [&](int mb, int t) {
          const int p = (rg * MBW + mb) * 16 + ln;
          return *reinterpret_cast<const f32x4 *>(cur + p * ROWB + 16 * ATileF<CIN>::pos(p, 4 * t + g));
        };
        auto group = [&](int t, const f32x4 (&a)[MBW]) {   // the MFMAs of channel group t, all blocks
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
              for (int i = 0; i < NBW; ++i)
                acc[mb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[i][t][j], a[mb][j], acc[mb][i], 0, 0, 0);
        };
        const bool full = bmw == (1 << MBW) - 1;
        f32x4 a[2][MBW];
        if (full) {   // first channel group of a full phase
#pragma unroll
          for (int mb = 0; mb < MBW; ++mb) a[0][mb] = a_at(mb, 0);
#pragma unroll
          for (int mb = 0; mb < MBW; ++mb) a[1][mb] = a_at(mb, 1);
          __builtin_amdgcn_sched_barrier(0);
          group(0, a[0]);
          __builtin_amdgcn_sched_barrier(0);
        }
        // ONE copy of the issue block, on every path (the ISA check walks all branches)
        issue();
        __builtin_amdgcn_sched_barrier(0);
        if (full) {
#pragma unroll
          for (int t = 1; t < T; ++t) {
            if (t + 1 < T) {
#pragma unroll
              for (int mb = 0; mb < MBW; ++mb) a[(t + 1) & 1][mb] = a_at(mb, t + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            group(t, a[t & 1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
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
                    acc[mb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[i][t][j], av[j], acc[mb][i],
                                                                     0, 0, 0);
              }
            }
        }
      }
      {   // the filter registers are free again: filters of phase q+1
        const int k1 = iB & 31;
#pragma unroll
        for (int i = 0; i < NBW; ++i)
#pragma unroll
          for (int t = 0; t < T; ++t)
            bf[i][t] = wp[(((size_t)k1 * NB + (cg * NBW + i)) * T + t) * 64 + lane];
      }
      if (fl & 4) {   // end of a tile: folded BN, residual, ReLU, straight from the accumulators
        f32x4 sc[NBW], sh[NBW], rsd[MBW][NBW];
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
          const int c = (cg * NBW + i) * 16 + 4 * g;
          sc[i] = scale ? *reinterpret_cast<const f32x4 *>(scale + c) : (f32x4){1.f, 1.f, 1.f, 1.f};
          sh[i] = scale ? *reinterpret_cast<const f32x4 *>(shift + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int mb = 0; mb < MBW; ++mb) {
            const int row = orow[mb] < 0 ? 0 : orow[mb];
            rsd[mb][i] = residual ? *reinterpret_cast<const f32x4 *>(residual + (size_t)row * COUT + c)
                                  : (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb) {
          const int row = orow[mb];
#pragma unroll
          for (int i = 0; i < NBW; ++i) {
            const int c = (cg * NBW + i) * 16 + 4 * g;
            f32x4 v = acc[mb][i];
            if (scale) v = v * sc[i] + sh[i];
            if (residual) v += rsd[mb][i];
            if (relu) {
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
            }
            if (row >= 0) *reinterpret_cast<f32x4 *>(out + (size_t)row * COUT + c) = v;
          }
        }
      }
      if ((fl & 8) && tid < kHdrInts && cj + 2 < cnt) s_hdr[((cj + 2) % 3) * kHdrInts + tid] = hreg;
      // end of the phase: this wave's share of the rows of phase q+1 has landed (vmcnt counts in
      // order), LDS writes are done, then the barrier -- no fence, the younger DMA stays in flight
      // (compiler fences around the pair: no memory access -- the LDS reads of the next phase
      // included -- moves across the counted wait and the raw barrier)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt((kWaitN & 15) | ((kWaitN >> 4) << 14) | (7 << 4) | (0 << 8));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    do {
      phase(std::integral_constant<int, 0>{});
      phase(std::integral_constant<int, 1>{});
      phase(std::integral_constant<int, 2>{});
    } while (((iA | iB | iC | iD) >> 13) & 1);
  }
  if (wg_times && tid == 0) wg_times[G + b] = wall_clock64();
}

unsigned long long *g_ord_prof = nullptr;
unsigned long long *g_dma_wg_times = nullptr;   // diagnostic only, see sf_debug_dma_wg_times   // diagnostic only, see sf_debug_ordered_profile
int g_ord_grid_cap = 0;                     // test hook, see sf_debug_ordered_grid_cap (0 = off)

struct OrdArgs {
  const float *feat;
  unsigned feat_bytes;
  int K, k_first;
  const int *plan;
  const float *wpk, *scale, *shift, *residual;
  int relu;
  float *out;
  const int *n_out_dev;
  int n_out_bound;
  hipStream_t st;
};

PlanView plan_view(const int *plan, int n_bound) {
  const int tiles = plan_tiles(n_bound);
  PlanView v;
  v.order = plan;
  v.sched = plan + n_bound;
  v.hdr = v.sched + tiles;
  v.nbrs = v.hdr + (size_t)tiles * kHdrInts;
  v.s_stride = tiles * kTileRows;
  return v;
}

template <int CIN, int COUT, int CG, int WAVES, int MINB, int BFD = 2>
int launch_ordered(const OrdArgs &a) {
  constexpr size_t lds = (size_t)2 * kTileRows * (CIN / 4) * 16 + (size_t)(3 * kHdrInts + 128) * 4;
  static_assert(lds <= 160 * 1024, "tile does not fit LDS");
  auto kern = conv_f32_ordered_kernel<CIN, COUT, CG, WAVES, MINB, BFD>;
  static SfSlotCache cache = {};  // resident workgroups per device
  const int slots = sf_resident_slots(cache, kern, WAVES * 64, lds);
  SF_CHECK_ARG(slots > 0, "cannot query the device for the resident workgroups");
  const int tiles = plan_tiles(a.n_out_bound);
  int grid = tiles < slots ? tiles : slots;
  if (g_ord_grid_cap > 0 && grid > g_ord_grid_cap) grid = g_ord_grid_cap;
  sf_note_kernel("conv_f32_ordered_kernel<%d, %d, %d, %d, %d, %d>", CIN, COUT, CG, WAVES, MINB, BFD);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, a.st, a.feat, a.feat_bytes,
                     plan_view(a.plan, a.n_out_bound), a.K, a.k_first,
                     reinterpret_cast<const f32x4 *>(a.wpk), a.scale, a.shift, a.residual, a.relu,
                     a.out, a.n_out_dev, a.n_out_bound, g_ord_prof);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

template <int CIN, int COUT, int CG, int WAVES, int MINB>
int launch_dma(const OrdArgs &a) {
  auto kern = conv_f32_dma_kernel<CIN, COUT, CG, WAVES, MINB>;
  static SfSlotCache cache = {};  // resident workgroups per device
  const int slots = sf_resident_slots(cache, kern, WAVES * 64, 0);
  SF_CHECK_ARG(slots > 0, "cannot query the device for the resident workgroups");
  const int tiles = plan_tiles(a.n_out_bound);
  int grid = tiles < slots ? tiles : slots;
  if (g_ord_grid_cap > 0 && grid > g_ord_grid_cap) grid = g_ord_grid_cap;
  sf_note_kernel("conv_f32_dma_kernel<%d, %d, %d, %d, %d>", CIN, COUT, CG, WAVES, MINB);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), 0, a.st, a.feat, a.feat_bytes,
                     plan_view(a.plan, a.n_out_bound), a.K, a.k_first,
                     reinterpret_cast<const f32x4 *>(a.wpk), a.scale, a.shift, a.residual, a.relu,
                     a.out, a.n_out_dev, a.n_out_bound, g_dma_wg_times);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// Wave grid per shape: the column groups CG divide COUT/16; 8 waves where a wave's filter
// registers (2 x NBW x CIN/16 float4) would otherwise not fit.
template <int CIN>
int dispatch_ordered(int cout, const OrdArgs &a) {
  switch (cout) {
    case 16: return launch_ordered<CIN, 16, 1, 4, 2>(a);
    case 32: return launch_ordered<CIN, 32, 2, 4, 2>(a);
    // The two square wide shapes run on the LDS-DMA kernel (64 -> 64: 136 vs 141 us for the
    // register-staged kernel with one filter set; 128 -> 128: 184 vs 192 us); the others keep the
    // register-staged kernel.
    case 64:
      if constexpr (CIN == 64) return launch_dma<CIN, 64, 4, 4, 3>(a);
      else return launch_ordered<CIN, 64, 4, 4, 2, 1>(a);
    case 128:
      if constexpr (CIN >= 128) return launch_dma<CIN, 128, 8, 8, 1>(a);
      else return launch_ordered<CIN, 128, 4, 4, 2>(a);
  }
  sf_set_error("unsupported cout=%d (16, 32, 64 or 128)", cout);
  return SF_ERR_BAD_ARG;
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
int sort_blocks(int n_bound) { return sf_div_up(n_bound > 0 ? n_bound : 1, kSortChunk); }

}  // namespace

extern "C" size_t sf_row_order_workspace(int n_bound) {
  if (n_bound < 1) n_bound = 1;
  const size_t arr = align_up((size_t)n_bound * 4);
  // keys x2, rows x2, three per-chunk histogram tables
  return 4 * arr + 3 * align_up((size_t)sort_blocks(n_bound) * kSortBins * 4);
}

extern "C" int sf_row_order_size(int n_bound) {
  if (n_bound < 1) n_bound = 1;
  const long long tiles = plan_tiles(n_bound);
  const long long ints = (long long)n_bound + tiles + tiles * kHdrInts + 27ll * tiles * kTileRows;
  return ints < 0x7FFFFFFFll ? (int)ints : -1;   // -1: more than 2^31 ints (n_bound beyond ~75 M rows)
}

extern "C" int sf_row_order_by_mask(const int *nbr, int nbr_stride, int K, int k_first, const int *n_dev,
                                    int n_bound, int *row_order, void *workspace,
                                    size_t workspace_bytes, void *stream) {
  return sf_int_row_order(nbr, nbr_stride, K, k_first, n_dev, n_bound, row_order, workspace, workspace_bytes,
                          0, (hipStream_t)stream);
}

// the three per-chunk histogram tables inside a row-order workspace (contiguous)
void sf_int_row_order_hist(void *workspace, int n_bound, uint32_t **hist, size_t *words) {
  if (n_bound < 1) n_bound = 1;
  const size_t arr = align_up((size_t)n_bound * 4);
  const size_t tab = align_up((size_t)sort_blocks(n_bound) * kSortBins * 4);
  *hist = (uint32_t *)((char *)workspace + 4 * arr);
  *words = 3 * tab / 4;
}

int sf_int_row_order(const int *nbr, int nbr_stride, int K, int k_first, const int *n_dev, int n_bound,
                     int *row_order, void *workspace, size_t workspace_bytes, int hist_ready,
                     hipStream_t st) {
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
  const size_t tab = align_up((size_t)nblk * kSortBins * 4);
  char *ws = (char *)workspace;
  unsigned *keys0 = (unsigned *)ws, *keys1 = (unsigned *)(ws + arr);
  int *rows0 = (int *)(ws + 2 * arr), *rows1 = (int *)(ws + 3 * arr);
  int *hist0 = (int *)(ws + 4 * arr), *hist1 = (int *)(ws + 4 * arr + tab),
      *hist2 = (int *)(ws + 4 * arr + 2 * tab);
  const PlanView pv = plan_view(row_order, n_bound);
  if (!hist_ready) {
    hipLaunchKernelGGL(hist_zero_kernel, dim3(nblk < 128 ? nblk : 128), dim3(256), 0, st, n_dev, n_bound,
                       hist0, tab / 4);
    SF_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(mask_hist_kernel, dim3(nblk * (kSortChunk / 256)), dim3(256), 0, st, nbr, nbr_stride, K, n_dev,
                     n_bound, keys0, hist0);
  SF_CHECK_LAUNCH();
  // three stable passes over the 27 mask bits; the last one lands in row_order
  hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblk), dim3(256), 0, st, (const unsigned *)keys0,
                     (const int *)nullptr, n_dev, n_bound, 0, (const int *)hist0, keys1, rows1, hist1);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblk), dim3(256), 0, st, (const unsigned *)keys1,
                     (const int *)rows1, n_dev, n_bound, 9, (const int *)hist1, keys0, rows0, hist2);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblk), dim3(256), 0, st, (const unsigned *)keys0,
                     (const int *)rows0, n_dev, n_bound, 18, (const int *)hist2, (unsigned *)nullptr,
                     row_order, (int *)nullptr);
  SF_CHECK_LAUNCH();
  const int tiles = plan_tiles(n_bound);
  hipLaunchKernelGGL(tile_plan_kernel, dim3(sf_grid(tiles, 4)), dim3(256), 0, st, nbr, nbr_stride, K,
                     k_first, n_dev, n_bound, row_order, const_cast<int *>(pv.hdr),
                     const_cast<int *>(pv.nbrs), pv.s_stride);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(tile_schedule_kernel, dim3(1), dim3(1024), 0, st, pv.hdr, n_dev, n_bound,
                     const_cast<int *>(pv.sched));
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
  SF_CHECK_ARG(row_order, "row_order is NULL: the kernel runs on the plan sf_row_order_by_mask wrote");
  SF_CHECK_ARG(((uintptr_t)features_in & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
                   ((uintptr_t)w_packed & 15) == 0 && ((uintptr_t)residual & 15) == 0 &&
                   ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0,
               "feature / filter / epilogue pointers must be 16-byte aligned");
  const long long bytes = (long long)n_in_bound * cin_padded * 4;
  SF_CHECK_ARG(n_in_bound >= 1 && bytes < (1ll << 32) - 4096,
               "the input feature table (%d rows) must stay below 4 GiB", n_in_bound);
  OrdArgs a{features_in, (unsigned)bytes, K, k_first, row_order, w_packed, scale, shift, residual,
            relu, out, n_out_dev, n_out_bound, (hipStream_t)stream};
  switch (cin_padded) {
    case 16: return dispatch_ordered<16>(cout, a);
    case 32: return dispatch_ordered<32>(cout, a);
    case 64: return dispatch_ordered<64>(cout, a);
    case 128: return dispatch_ordered<128>(cout, a);
  }
  sf_set_error("unsupported cin_padded=%d for the ordered kernel (16, 32, 64 or 128)", cin_padded);
  return SF_ERR_BAD_ARG;
}

// Diagnostic hook (not part of the documented ABI): a device buffer of AT LEAST 16 uint64 (nine are
// written today) that every register-staged ordered conv launch accumulates per-wave stamps into:
// [0] issue, [1] multiply, [2] epilogue, [3] barrier cycles, [4] phases, [5] multiplying phases,
// [6] cycles in kernel, [7] waves, [8] 100 MHz ticks in kernel; NULL = off.
extern "C" __attribute__((visibility("default"))) void sf_debug_ordered_profile(void *dev_buf16) {
  g_ord_prof = (unsigned long long *)dev_buf16;
}

// Diagnostic hook (not part of the documented ABI): a device buffer of 2 x 4096 uint64; every LDS-DMA
// conv launch writes the 100 MHz wall clock at the start ([b]) and end ([grid + b]) of workgroup b
// (tools/prof_dma_wg.py: how level the persistent workgroups finish).  NULL = off.
extern "C" __attribute__((visibility("default"))) void sf_debug_dma_wg_times(void *dev_buf) {
  g_dma_wg_times = (unsigned long long *)dev_buf;
}

// Test hook (not part of the documented ABI): limits the grid of the ordered / DMA kernels to `cap`
// persistent workgroups (0 = off), so that a small problem gives a workgroup more than kRun tiles
// (tests/test_gpu_conv_ordered.py::test_many_tiles_per_workgroup).  Process-wide, not thread-safe.
extern "C" __attribute__((visibility("default"))) void sf_debug_ordered_grid_cap(int cap) {
  g_ord_grid_cap = cap > 0 ? cap : 0;
}

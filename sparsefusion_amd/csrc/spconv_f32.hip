// Output-stationary sparse convolution (fp32, exact-f32 MFMA) for gfx950.
//
// Reference: indiceConv<float> (mmdet3d/ops/spconv/include/spconv/spconv_ops.h:260-361) runs,
// per kernel offset, gather -> torch::mm -> scatter-add through two HBM staging buffers
// (52-54 launches per layer).  Here one workgroup owns a tile of 64 OUTPUT rows and keeps it in
// LDS for the whole layer; per offset it compacts the tile's active (input,output) pairs
// (wave ballot on the output-major neighbour table), gathers exactly those input rows into LDS
// (16-byte coalesced row pieces), multiplies them by W[k] on the matrix cores
// (v_mfma_f32_16x16x4_f32: bit-exact fp32 FMA chains, no zero rows multiplied) and adds the
// products into the LDS-resident output tile.  Each output row is written to HBM once, with the
// folded BatchNorm / residual / ReLU epilogue applied on the way out; no atomics, no staging
// buffers, deterministic summation order (offsets ascending, centre first for SubM as the
// reference does).
//
// Algorithmic HBM bytes per layer:
//   P*Cin*4 (gather) + N_out*Cout*4*(1+residual) (single write [+identity read]) + P*8 (indices)
#include <stdlib.h>

#include <type_traits>

#include "spconv_internal.h"

namespace {

constexpr int kMaxK = 27;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int load_n(const int *n_dev, int n_bound) {
  int n = n_dev ? *n_dev : n_bound;
  return n < n_bound ? n : n_bound;
}

template <int CIN>
struct ATile {
  static constexpr int CH = CIN / 4;                       // 16-byte chunks per row
  static constexpr int R = (CH >= 16) ? 1 : 16 / CH;        // rows per 256-B LDS bank row
  static constexpr int M = (CH >= 16) ? 15 : CH - 1;
  // XOR swizzle of the chunk position so the 16 rows one MFMA fragment read touches spread
  // over the 64 banks (guide: Guideline 4 / T2); the same function is used to write and read.
  __device__ static __forceinline__ int pos(int p, int c) { return c ^ ((p / R) & M); }
};

// One 16-row block (or two at once: two independent accumulator chains keep the fp32 MFMA pipe
// at its 32-cycle issue rate instead of the 40-cycle dependent latency) of compacted rows times
// the wave's filter columns; the products are added into the LDS-resident output tile (plain
// read-add-write: rows are unique per offset and offsets are separated by barriers; LDS float
// atomics measured 2x slower end to end).
//   * the four destination rows of a lane are four consecutive bytes of the offset's list: one
//     ds_read_b32 (list padding entries point at the spare row TM, so no bounds test);
//   * the tile values to update are read BEFORE the MFMA chain so their LDS latency hides
//     under it; A fragments are read one channel group ahead of the MFMAs that use them.
template <int CIN, int NBW, int T, int NS, int OST>
__device__ __forceinline__ void mma_blocks(const float *__restrict__ sa, float *__restrict__ s_out,
                                           const unsigned char *__restrict__ dstk,
                                           const f32x4 (&b)[NBW][T], const int (&sbs)[NS], int col0,
                                           int g, int ln) {
  constexpr int CH = CIN / 4;
  // NS == 1: split the channel groups over two accumulators (even / odd t) so the single block
  // still runs two independent MFMA chains; summed once at the end.
  constexpr int NA = (NS == 1 && T >= 2) ? 2 : NS;
  f32x4 acc[NA][NBW];
  f32x4 a_cur[NS], a_nxt[NS];
  unsigned d4[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    d4[s] = *reinterpret_cast<const unsigned *>(dstk + sbs[s] * 16 + g * 4);
    const int p = sbs[s] * 16 + ln;
    a_cur[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (CIN >= 16 || g < CH)
      a_cur[s] = *reinterpret_cast<const f32x4 *>(&sa[p * CIN + 4 * ATile<CIN>::pos(p, g)]);
  }
#pragma unroll
  for (int s = 0; s < NA; ++s)
#pragma unroll
    for (int i = 0; i < NBW; ++i) acc[s][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int off[NS][4];
  float cur[NS][4][NBW];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      off[s][r] = (int)((d4[s] >> (8 * r)) & 255u) * OST + col0 + ln;
#pragma unroll
      for (int i = 0; i < NBW; ++i) cur[s][r][i] = s_out[off[s][r] + i * 16];
    }
  __builtin_amdgcn_sched_barrier(0);  // keep these LDS reads ahead of the MFMA chain
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (t + 1 < T) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int p = sbs[s] * 16 + ln;
        a_nxt[s] = *reinterpret_cast<const f32x4 *>(
            &sa[p * CIN + 4 * ATile<CIN>::pos(p, 4 * (t + 1) + g)]);
      }
      __builtin_amdgcn_sched_barrier(0);  // issue the next group's reads before this group's MFMAs
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const int a = (NS == 1 && NA == 2) ? (t & 1) : s;
          acc[a][i] =
              __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s][j], b[i][t][j], acc[a][i], 0, 0, 0);
        }
    if (t + 1 < T) {
#pragma unroll
      for (int s = 0; s < NS; ++s) a_cur[s] = a_nxt[s];
    }
  }
  if (NS == 1 && NA == 2) {
#pragma unroll
    for (int i = 0; i < NBW; ++i) acc[0][i] += acc[1][i];
  }
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NBW; ++i) s_out[off[s][r] + i * 16] = cur[s][r][i] + acc[s][i][r];
}

// TM    output rows a tile can hold (the LDS-resident fp32 tile); the kernel sizes its tiles at
//       run time (<= TM) so that every workgroup of the grid gets the same number of tiles
// AM    rows of the gathered-input tile; an offset with more than AM pairs in the tile is
//       processed in chunks of AM (so TM is bounded by the output tile only)
// WAVES wavefronts per workgroup (4 or 8)
// ABUF  LDS buffers for the gathered rows (2: one barrier per phase, 1: two barriers, less LDS)
template <int CIN, int COUT, int TM, int AM, int WAVES, int ABUF>
__global__ void __launch_bounds__(WAVES * 64)
sparse_conv_f32_kernel(const float *__restrict__ feat, const int *__restrict__ nbr, int nbr_stride,
                       int K, int k_first, const float *__restrict__ wpk,
                       const float *__restrict__ scale, const float *__restrict__ shift,
                       const float *__restrict__ residual, int relu, float *__restrict__ out,
                       const int *n_out_dev, int n_out_bound, int balance,
                       unsigned long long *__restrict__ prof, const int *__restrict__ order) {
  constexpr int NT = WAVES * 64;
  constexpr int CINP = CIN < 16 ? 16 : CIN;  // packed-weight K extent
  constexpr int T = CINP / 16;               // 16-channel groups = float4 B fragments per N block
  constexpr int CH = CIN / 4;
  constexpr int NB = COUT / 16;
  constexpr int CG = NB < WAVES ? NB : WAVES;  // wave groups along output columns
  constexpr int NBW = NB / CG;                 // N blocks per wave
  constexpr int RG = WAVES / CG;               // wave groups along compact rows
  constexpr int OST = COUT + 4;                // padded output-tile row stride (floats)
  constexpr int TMP = (TM + 15) & ~15;         // list length: padded to whole 16-row blocks
  constexpr int GI = (AM * CH + NT - 1) / NT;  // gather items per thread
  constexpr int HALVES = (TM + 63) / 64;
  constexpr bool EXACT = (AM * CH) % NT == 0;  // every thread owns GI whole gather items
  constexpr int KPW = (kMaxK + WAVES - 1) / WAVES;  // offsets compacted per wave
  constexpr int NCH = (TM + AM - 1) / AM;           // chunks per offset at most
  static_assert(TM <= 255 && AM % 16 == 0 && TM % 4 == 0, "tile shape");

  __shared__ __attribute__((aligned(16))) float s_out[(TM + 1) * OST];  // + spare row
  __shared__ __attribute__((aligned(16))) float s_a[ABUF][AM * CIN];
  __shared__ int s_src[kMaxK][TMP];
  __shared__ __attribute__((aligned(4))) unsigned char s_dst[kMaxK][TMP];
  __shared__ int s_cnt[kMaxK];
  __shared__ int s_order[kMaxK * NCH + 1];  // phase list: k | (first pair of the chunk << 8)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = wave % CG, rg = wave / CG;
  const int g = lane >> 4, ln = lane & 15;
  const int n_out = load_n(n_out_dev, n_out_bound);
  const f32x4 *__restrict__ wp4 = reinterpret_cast<const f32x4 *>(wpk);

  // tile height: TM rows, or (balance) the largest height <= TM that gives every workgroup of
  // the grid the same number of tiles
  const int G = gridDim.x;
  int tm = TM;
  if (balance) {
    const int rounds = (n_out + G * TM - 1) / (G * TM);
    tm = rounds > 0 ? (n_out + G * rounds - 1) / (G * rounds) : TM;
    if (tm < 32) tm = 32;
    if (tm > TM) tm = TM;
  }
  const int tiles = (n_out + tm - 1) / tm;
  // diagnostic stamps (prof != NULL only in profiling runs): cycles of wave 0 per region
  unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tk0 = 0, tr0 = 0;
#define SF_STAMP() (prof ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull)
  if (prof) {
    tk0 = __builtin_amdgcn_s_memtime();
    tr0 = __builtin_amdgcn_s_memrealtime();
  }

  for (int tile = blockIdx.x; tile < tiles; tile += G) {
    unsigned long long ts = SF_STAMP();
    const int row0 = tile * tm;
    const int row_end = (row0 + tm < n_out) ? row0 + tm : n_out;
    // ---- setup: compact the tile's pairs per offset, clear the output tile --------------
    int nv[KPW][HALVES];
    int orow[HALVES];  // output row of tile entry h*64+lane (identity without a row order)
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
      const int o = row0 + h * 64 + lane;
      orow[h] = o < row_end ? (order ? order[o] : o) : -1;
    }
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      const int k = wave + i * WAVES;
#pragma unroll
      for (int h = 0; h < HALVES; ++h)
        nv[i][h] = (k < K && orow[h] >= 0) ? nbr[(size_t)k * nbr_stride + orow[h]] : -1;
    }
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      const int k = wave + i * WAVES;
      if (k < K) {
        int base = 0;
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
          int v = nv[i][h];
          unsigned long long m = __ballot(v >= 0);
          int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          if (v >= 0) {
            s_src[k][pos] = v;
            s_dst[k][pos] = (unsigned char)(h * 64 + lane);
          }
          base += __popcll(m);
        }
        // padding entries of the last 16-row block land in the spare row TM of the tile
        if (lane < ((16 - base) & 15)) s_dst[k][base + lane] = (unsigned char)TM;
        if (lane == 0) s_cnt[k] = base;
      }
    }
    for (int i = tid; i < TM * OST; i += NT) s_out[i] = 0.0f;
    __syncthreads();
    if (wave == 0) {
      // phase list: k_first (SubM centre) first, then k ascending; an offset with c pairs takes
      // ceil(c / AM) phases; empty offsets take none.  Lane k places offset k's phases.
      const int c = (lane < K) ? s_cnt[lane] : 0;
      const int nch = (c + AM - 1) / AM;
      const int incl = sf_wave_incl_scan(nch);
      const int total = __shfl(incl, 63, 64);
      int pre = incl - nch;
      if (k_first >= 0 && k_first < K) {
        const int cf = __shfl(nch, k_first, 64);
        pre = (lane < k_first) ? pre + cf : (lane == k_first ? 0 : pre);
      }
      for (int j = 0; j < nch; ++j) s_order[pre + j] = lane | ((j * AM) << 8);
      if (lane == 0) s_order[kMaxK * NCH] = total;
    }
    __syncthreads();
    const int n_order = s_order[kMaxK * NCH];

    // ---- phase pipeline ------------------------------------------------------------------
    // While phase e multiplies the rows parked in LDS, the rows of phase e+1 are in flight
    // into register set (e+1)&1 (issued one phase ago) and the loads of phase e+2 are issued
    // into set e&1: a random HBM row takes longer than one phase to arrive.  Phases past the
    // end re-read one cached row (count 0) so that every phase is the same straight-line code
    // and the compiler's load counters stay exact.
    f32x4 greg[2][GI];
    f32x4 bcur[NBW][T], bnext[NBW][T];
    auto phase_at = [&](int e, int &k, int &st, int &cnt) {
      const int ec = e < 0 ? 0 : (e < n_order ? e : n_order - 1);
      const int ent = s_order[ec];
      k = ent & 255;
      st = ent >> 8;
      int c = s_cnt[k] - st;
      c = c < AM ? c : AM;
      cnt = (e == ec) ? c : 0;
    };
    auto load_filters = [&](f32x4(&bf)[NBW][T], int k) {
#pragma unroll
      for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int t = 0; t < T; ++t)
          bf[i][t] = wp4[(((size_t)k * NB + (cg * NBW + i)) * T + t) * 64 + lane];
    };
    auto gather_issue = [&](f32x4(&gr)[GI], int k, int st, int cnt) {
#pragma unroll
      for (int m = 0; m < GI; ++m) {
        // rows past cnt are never consumed: they re-read list entry 0 (one cache line)
        int idx = tid + m * NT, p = idx / CH, c = idx % CH;
        if (EXACT || idx < AM * CH)
          gr[m] = *reinterpret_cast<const f32x4 *>(
              feat + (size_t)s_src[k][st + (p < cnt ? p : 0)] * CIN + 4 * c);
      }
    };
    auto gather_park = [&](const f32x4(&gr)[GI], float *sa) {
#pragma unroll
      for (int m = 0; m < GI; ++m) {
        int idx = tid + m * NT, p = idx / CH, c = idx % CH;
        if (EXACT || idx < AM * CH)
          *reinterpret_cast<f32x4 *>(&sa[p * CIN + 4 * ATile<CIN>::pos(p, c)]) = gr[m];
      }
    };
    // one phase; PAR = parity of kk = register set that receives the loads of phase kk+2
    auto phase = [&](int kk, auto PAR) {
      constexpr int par = decltype(PAR)::value;
      unsigned long long t0 = SF_STAMP();
      int k, st, cnt, k1, st1, cnt1, k2, st2, cnt2;
      phase_at(kk, k, st, cnt);
      phase_at(kk + 1, k1, st1, cnt1);
      phase_at(kk + 2, k2, st2, cnt2);
      float *sa_cur = s_a[ABUF == 2 ? par : 0];
      float *sa_nxt = s_a[ABUF == 2 ? (par ^ 1) : 0];
      // (1) issue: filters of phase kk+1, rows of phase kk+2
      load_filters(bnext, k1);
      gather_issue(greg[par], k2, st2, cnt2);
      unsigned long long t1 = SF_STAMP();
      // (2) this phase: MFMA over the compacted rows, products added into the LDS tile
      const int nsb = (cnt + 15) >> 4;
      const unsigned char *dstk = &s_dst[k][st];
      int sb = rg;
      for (; sb + RG < nsb; sb += 2 * RG) {
        const int sbs[2] = {sb, sb + RG};
        mma_blocks<CIN, NBW, T, 2, OST>(sa_cur, s_out, dstk, bcur, sbs, cg * NBW * 16, g, ln);
      }
      if (sb < nsb) {
        const int sbs[1] = {sb};
        mma_blocks<CIN, NBW, T, 1, OST>(sa_cur, s_out, dstk, bcur, sbs, cg * NBW * 16, g, ln);
      }
      unsigned long long t2 = SF_STAMP();
      // (3) park the rows of phase kk+1 (issued one phase ago) in LDS
      if (ABUF == 1) __syncthreads();
      unsigned long long t3 = SF_STAMP();
      gather_park(greg[par ^ 1], sa_nxt);
#pragma unroll
      for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int t = 0; t < T; ++t) bcur[i][t] = bnext[i][t];
      unsigned long long t4 = SF_STAMP();
      __syncthreads();
      if (prof) {
        unsigned long long t5 = __builtin_amdgcn_s_memtime();
        pt[1] += t1 - t0;  // issue next gather + filter loads
        pt[2] += t2 - t1;  // MFMA + tile update
        pt[3] += t3 - t2;  // barrier 1
        pt[4] += t4 - t3;  // wait loads, LDS store
        pt[5] += t5 - t4;  // barrier 2
        pt[7] += 1;
      }
    };
    {
      unsigned long long te = SF_STAMP();
      pt[0] += te - ts;  // setup
    }
    // The pipeline is primed by the loop itself: phases -2 and -1 have no rows to multiply,
    // they only issue the loads of phases 0 and 1.  One loop body, one back edge, nothing
    // carried into the loop: the compiler's outstanding-load counts stay exact.
    if (n_order > 0) {
#pragma unroll
      for (int m = 0; m < GI; ++m) greg[0][m] = greg[1][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int t = 0; t < T; ++t) bcur[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int kk = -2; kk < n_order; kk += 2) {
        phase(kk, std::integral_constant<int, 0>{});
        phase(kk + 1, std::integral_constant<int, 1>{});
      }
    }
    unsigned long long tq = SF_STAMP();

    // ---- epilogue: folded BN, residual, ReLU; one coalesced write per output row ---------
    constexpr int OC = COUT / 4;
    for (int idx = tid; idx < TM * OC; idx += NT) {
      int r = idx / OC, c = idx % OC, row = row0 + r;
      if (row >= row_end) continue;
      if (order) row = order[row];
      f32x4 v = *reinterpret_cast<const f32x4 *>(&s_out[r * OST + 4 * c]);
      if (scale) {
        f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + 4 * c);
        f32x4 sh = *reinterpret_cast<const f32x4 *>(shift + 4 * c);
        v = v * sc + sh;
      }
      if (residual) v += *reinterpret_cast<const f32x4 *>(residual + (size_t)row * COUT + 4 * c);
      if (relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
      }
      *reinterpret_cast<f32x4 *>(out + (size_t)row * COUT + 4 * c) = v;
    }
    __syncthreads();  // s_out / lists are reused by the next tile
    if (prof) pt[6] += (unsigned long long)__builtin_amdgcn_s_memtime() - tq;  // epilogue
  }
  if (prof && threadIdx.x == 0) {
    unsigned long long tk1 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 8; ++i) atomicAdd(&prof[i], pt[i]);
    atomicAdd(&prof[8], tk1 - tk0);   // cycles in kernel (this workgroup)
    atomicAdd(&prof[9], tr1 - tr0);   // 100 MHz ticks in kernel
    atomicAdd(&prof[10], 1ull);       // workgroups
  }
#undef SF_STAMP
}

// ---------------------------------------------------------------------------------------------
// "Wide" variant: Cout/16 is a multiple of the wave count, so every wave owns its output columns
// for ALL compacted rows of a phase.  Nobody else touches a wave's columns of the LDS tile, a
// phase needs ONE barrier (the gathered rows are double buffered), and the phase body is
// straight-line code specialised on the number of 16-row blocks.  The bookkeeping of the pipeline
// (list reads, address arithmetic, loads for later phases, parking gathered rows, the
// read-add-write of finished blocks) is spread over the MFMA stream with sched_group_barrier, so
// it issues in the shadow of the matrix pipe instead of in front of it.
template <int CIN, int COUT, int TM, int AM, int WAVES>
struct WideShape {
  static constexpr int NT = WAVES * 64;
  static constexpr int T = CIN / 16;    // 16-channel groups
  static constexpr int CH = CIN / 4;    // 16-byte chunks per row
  static constexpr int NB = COUT / 16;  // N blocks
  static constexpr int NBW = NB / WAVES;
  static constexpr int OST = COUT + 4;
  static constexpr int TMP = (TM + 15) & ~15;
  static constexpr int GI = AM * CH / NT;
  static constexpr int HALVES = (TM + 63) / 64;
  static constexpr int KPW = (kMaxK + WAVES - 1) / WAVES;
  static constexpr int NCH = (TM + AM - 1) / AM;
  static constexpr int NMETA = kMaxK * NCH + 6;
  static constexpr int MAXSB = AM / 16;
  static constexpr int cin = CIN;
  static_assert(NB % WAVES == 0 && (AM * CH) % NT == 0 && T >= 2 && AM % 16 == 0 && AM <= 64 &&
                    TM <= 255 && TM % 4 == 0,
                "shape not supported by the wide kernel");
};

// phase descriptor: offset k | first pair of the chunk << 8 | pairs in the chunk << 16
#define SF_META_K(m) ((m) & 31)
#define SF_META_ST(m) (((m) >> 8) & 255)
#define SF_META_CNT(m) (((m) >> 16) & 255)

template <class S, int NSB, int PAR>
__device__ __forceinline__ void wide_phase(const float *__restrict__ feat,
                                           const f32x4 *__restrict__ wp4, float *__restrict__ s_out,
                                           float *__restrict__ sa_cur, float *__restrict__ sa_nxt,
                                           const int *__restrict__ s_src,
                                           const unsigned char *__restrict__ s_dst,
                                           const int *__restrict__ meta_i, int tid, int lane,
                                           int wave, int &m0, int &m1, f32x4 (&greg)[2][S::GI],
                                           f32x4 (&bf)[2][S::NBW][S::T]) {
  constexpr int T = S::T, NBW = S::NBW, GI = S::GI, CH = S::CH, OST = S::OST, TMP = S::TMP;
  constexpr int NT = S::NT, NB = S::NB, CIN = S::cin;
  constexpr int NG = (NSB + 1) / 2;  // groups of two row blocks (the last may hold one)
  const int g = lane >> 4, ln = lane & 15, col0 = wave * NBW * 16;
  const int k0 = SF_META_K(m0), st0 = SF_META_ST(m0), k1 = SF_META_K(m1);
  const unsigned char *dstk = s_dst + k0 * TMP + st0;

  int m2v = 0, m2 = 0;
  int srow[GI];
  unsigned d4[NG > 0 ? NG : 1][2];
  f32x4 a_cur[2], a_nxt[2];
  f32x4 acc[2][NBW], accp[2][NBW];
  int off[2][4], offp[2][4];
  float cur[2][4][NBW], curp[2][4][NBW];

  // ---- pipeline bookkeeping, one small piece per MFMA unit -----------------------------
  auto side = [&](int s) {
    if (s == 0) {  // descriptor of phase i+2; filter fragments of phase i+1
      m2v = meta_i[2];
#pragma unroll
      for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int t = 0; t < T; ++t)
          bf[PAR ^ 1][i][t] = wp4[(((size_t)k1 * NB + (wave * NBW + i)) * T + t) * 64 + lane];
    } else if (s == 1) {  // input rows of phase i+2 (past the count: entry 0, one cached row)
      m2 = __builtin_amdgcn_readfirstlane(m2v);
      const int k2 = SF_META_K(m2), st2 = SF_META_ST(m2), cnt2 = SF_META_CNT(m2);
#pragma unroll
      for (int m = 0; m < GI; ++m) {
        const int p = (tid + m * NT) / CH;
        srow[m] = s_src[k2 * TMP + st2 + (p < cnt2 ? p : 0)];
      }
    } else if (s == 2) {  // gather loads of phase i+2
#pragma unroll
      for (int m = 0; m < GI; ++m) {
        const int c = (tid + m * NT) % CH;
        greg[PAR][m] = *reinterpret_cast<const f32x4 *>(feat + (size_t)srow[m] * CIN + 4 * c);
      }
    } else if (s == 3) {  // park the rows of phase i+1 (issued one phase ago)
#pragma unroll
      for (int m = 0; m < GI; ++m) {
        const int idx = tid + m * NT, p = idx / CH, c = idx % CH;
        *reinterpret_cast<f32x4 *>(&sa_nxt[p * CIN + 4 * ATile<CIN>::pos(p, c)]) = greg[PAR ^ 1][m];
      }
    }
  };
  auto a_read = [&](f32x4(&a)[2], int gi, int t, int ns) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (s < ns) {
        const int p = (2 * gi + s) * 16 + ln;
        a[s] = *reinterpret_cast<const f32x4 *>(&sa_cur[p * CIN + 4 * ATile<CIN>::pos(p, 4 * t + g)]);
      }
  };
  auto scatter_prev = [&](int ns) {
    if (ns == 1) {  // a lone block ran as two chains (even / odd channel groups)
#pragma unroll
      for (int i = 0; i < NBW; ++i) accp[0][i] += accp[1][i];
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (s < ns) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < NBW; ++i)
            s_out[offp[s][r] + i * 16] = curp[s][r][i] + accp[s][i][r];
      }
  };

  if (NG > 0) {
#pragma unroll
    for (int gi = 0; gi < NG; ++gi)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        if (2 * gi + s < NSB)
          d4[gi][s] = *reinterpret_cast<const unsigned *>(dstk + (2 * gi + s) * 16 + g * 4);
    a_read(a_cur, 0, 0, NSB >= 2 ? 2 : 1);
  }
  int u = 0;
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int ns = (2 * gi + 1 < NSB) ? 2 : 1;
#pragma unroll
    for (int t = 0; t < T; ++t, ++u) {
      if (u < 4) side(u);
      if (t == 0) {
        if (gi > 0) scatter_prev(2);  // every group before the last holds two blocks
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int i = 0; i < NBW; ++i) acc[s][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (s < ns) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              off[s][r] = (int)((d4[gi][s] >> (8 * r)) & 255u) * OST + col0 + ln;
#pragma unroll
              for (int i = 0; i < NBW; ++i) cur[s][r][i] = s_out[off[s][r] + i * 16];
            }
          }
        }
      }
      if (t + 1 < T)
        a_read(a_nxt, gi, t + 1, ns);
      else if (gi + 1 < NG)
        a_read(a_nxt, gi + 1, 0, (2 * gi + 3 < NSB) ? 2 : 1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < NBW; ++i)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            if (s < ns) {
              const int a = (ns == 2) ? s : (t & 1);
              acc[a][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s][j], bf[PAR][i][t][j],
                                                              acc[a][i], 0, 0, 0);
            }
      // one MFMA, then a few of everything else: the bookkeeping issues while the pipe is busy
#pragma unroll
      for (int q = 0; q < 4 * NBW * 2; ++q) {
        if (q < 4 * NBW * ns) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x096, 6, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 2; ++s) a_cur[s] = a_nxt[s];
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < NBW; ++i) accp[s][i] = acc[s][i];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        offp[s][r] = off[s][r];
#pragma unroll
        for (int i = 0; i < NBW; ++i) curp[s][r][i] = cur[s][r][i];
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (s >= u) side(s);
  if (NG > 0) scatter_prev((NSB & 1) ? 1 : 2);
  m0 = m1;
  m1 = m2;
}

template <int CIN, int COUT, int TM, int AM, int WAVES, int MINW>
__global__ void __launch_bounds__(WAVES * 64, MINW)
sparse_conv_f32_wide_kernel(const float *__restrict__ feat, const int *__restrict__ nbr,
                            int nbr_stride, int K, int k_first, const float *__restrict__ wpk,
                            const float *__restrict__ scale, const float *__restrict__ shift,
                            const float *__restrict__ residual, int relu, float *__restrict__ out,
                            const int *n_out_dev, int n_out_bound, int balance,
                            unsigned long long *__restrict__ prof,
                            const int *__restrict__ order) {
  using S = WideShape<CIN, COUT, TM, AM, WAVES>;
  constexpr int NT = S::NT, T = S::T, NBW = S::NBW, OST = S::OST, TMP = S::TMP, GI = S::GI;
  constexpr int HALVES = S::HALVES, KPW = S::KPW, NMETA = S::NMETA;

  __shared__ __attribute__((aligned(16))) float s_out[(TM + 1) * OST];  // + spare row
  __shared__ __attribute__((aligned(16))) float s_a[2][AM * CIN];
  __shared__ int s_src[kMaxK * TMP];
  __shared__ __attribute__((aligned(4))) unsigned char s_dst[kMaxK * TMP];
  __shared__ int s_cnt[kMaxK];
  __shared__ int s_meta[NMETA + 1];  // [NMETA] = number of phases

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_out = load_n(n_out_dev, n_out_bound);
  const f32x4 *__restrict__ wp4 = reinterpret_cast<const f32x4 *>(wpk);

  const int G = gridDim.x;
  int tm = TM;
  if (balance) {
    const int rounds = (n_out + G * TM - 1) / (G * TM);
    tm = rounds > 0 ? (n_out + G * rounds - 1) / (G * rounds) : TM;
    if (tm < 32) tm = 32;
    if (tm > TM) tm = TM;
  }
  const int tiles = (n_out + tm - 1) / tm;
  // diagnostic stamps (prof != NULL only in profiling runs): cycles per region, per tile
  unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tk0 = 0, tr0 = 0;
#define SF_STAMP() (prof ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull)
  if (prof) {
    tk0 = __builtin_amdgcn_s_memtime();
    tr0 = __builtin_amdgcn_s_memrealtime();
  }

  for (int tile = blockIdx.x; tile < tiles; tile += G) {
    unsigned long long ts = SF_STAMP();
    const int row0 = tile * tm;
    const int row_end = (row0 + tm < n_out) ? row0 + tm : n_out;
    // ---- setup: compact the tile's pairs per offset, clear the output tile --------------
    int nv[KPW][HALVES];
    int orow[HALVES];  // output row of tile entry h*64+lane (identity without a row order)
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
      const int o = row0 + h * 64 + lane;
      orow[h] = o < row_end ? (order ? order[o] : o) : -1;
    }
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      const int k = wave + i * WAVES;
#pragma unroll
      for (int h = 0; h < HALVES; ++h)
        nv[i][h] = (k < K && orow[h] >= 0) ? nbr[(size_t)k * nbr_stride + orow[h]] : -1;
    }
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      const int k = wave + i * WAVES;
      if (k < K) {
        int base = 0;
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
          int v = nv[i][h];
          unsigned long long m = __ballot(v >= 0);
          int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          if (v >= 0) {
            s_src[k * TMP + pos] = v;
            s_dst[k * TMP + pos] = (unsigned char)(h * 64 + lane);
          }
          base += __popcll(m);
        }
        // padding entries of the last 16-row block land in the spare row TM of the tile
        if (lane < ((16 - base) & 15)) s_dst[k * TMP + base + lane] = (unsigned char)TM;
        if (lane == 0) s_cnt[k] = base;
      }
    }
    for (int i = tid; i < TM * OST; i += NT) s_out[i] = 0.0f;
    __syncthreads();
    if (wave == 0) {
      // phase list: k_first (SubM centre) first, then k ascending; an offset with c pairs takes
      // ceil(c / AM) phases; empty offsets take none.  Lane k places offset k's phases.  Two
      // empty phases in front prime the pipeline, four behind drain it.
      const int c = (lane < K) ? s_cnt[lane] : 0;
      const int nch = (c + AM - 1) / AM;
      const int incl = sf_wave_incl_scan(nch);
      const int total = __shfl(incl, 63, 64);
      int pre = incl - nch;
      if (k_first >= 0 && k_first < K) {
        const int cf = __shfl(nch, k_first, 64);
        pre = (lane < k_first) ? pre + cf : (lane == k_first ? 0 : pre);
      }
      for (int j = 0; j < nch; ++j) {
        const int left = c - j * AM;
        s_meta[2 + pre + j] = lane | ((j * AM) << 8) | ((left < AM ? left : AM) << 16);
      }
      const unsigned long long any = __ballot(nch > 0);
      const int kd = any ? (int)__ffsll((long long)any) - 1 : 0;  // an offset that has pairs
      if (lane < 2) s_meta[lane] = kd;
      if (lane >= 2 && lane < 6) s_meta[total + lane] = kd;
      if (lane == 0) s_meta[NMETA] = total;
    }
    __syncthreads();
    const int n_phase = s_meta[NMETA];
    unsigned long long tp = SF_STAMP();
    pt[0] += tp - ts;  // setup
    pt[7] += n_phase;

    if (n_phase > 0) {
      f32x4 greg[2][GI];
      f32x4 bf[2][NBW][T];
#pragma unroll
      for (int m = 0; m < GI; ++m) greg[0][m] = greg[1][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int t = 0; t < T; ++t) bf[0][i][t] = bf[1][i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      int m0 = __builtin_amdgcn_readfirstlane(s_meta[0]);
      int m1 = __builtin_amdgcn_readfirstlane(s_meta[1]);
      const int n_steps = (n_phase + 3) & ~1;  // two priming phases in front, rounded up to even
#define SF_WIDE_PHASE(NSB, PAR, I)                                                              \
  wide_phase<S, NSB, PAR>(feat, wp4, s_out, s_a[PAR], s_a[PAR ^ 1], s_src, s_dst, &s_meta[I], tid, \
                          lane, wave, m0, m1, greg, bf)
#define SF_WIDE_DISPATCH(PAR, I)                                   \
  switch ((SF_META_CNT(m0) + 15) >> 4) {                           \
    case 0: SF_WIDE_PHASE(0, PAR, I); break;                       \
    case 1: SF_WIDE_PHASE(1, PAR, I); break;                       \
    case 2: SF_WIDE_PHASE(2, PAR, I); break;                       \
    case 3: SF_WIDE_PHASE(3, PAR, I); break;                       \
    default:                                                       \
      if constexpr (S::MAXSB >= 4) SF_WIDE_PHASE(4, PAR, I);       \
      break;                                                       \
  }
      for (int i = 0; i < n_steps; i += 2) {
        SF_WIDE_DISPATCH(0, i)
        __syncthreads();
        SF_WIDE_DISPATCH(1, i + 1)
        __syncthreads();
      }
#undef SF_WIDE_DISPATCH
#undef SF_WIDE_PHASE
    }

    unsigned long long tq = SF_STAMP();
    pt[2] += tq - tp;  // all phases of the tile (incl. the two priming ones)
    // ---- epilogue: folded BN, residual, ReLU; one coalesced write per output row ---------
    constexpr int OC = COUT / 4;
    for (int idx = tid; idx < TM * OC; idx += NT) {
      int r = idx / OC, c = idx % OC, row = row0 + r;
      if (row >= row_end) continue;
      if (order) row = order[row];
      f32x4 v = *reinterpret_cast<const f32x4 *>(&s_out[r * OST + 4 * c]);
      if (scale) {
        f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + 4 * c);
        f32x4 sh = *reinterpret_cast<const f32x4 *>(shift + 4 * c);
        v = v * sc + sh;
      }
      if (residual) v += *reinterpret_cast<const f32x4 *>(residual + (size_t)row * COUT + 4 * c);
      if (relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
      }
      *reinterpret_cast<f32x4 *>(out + (size_t)row * COUT + 4 * c) = v;
    }
    __syncthreads();  // s_out / lists are reused by the next tile
    if (prof) pt[6] += (unsigned long long)__builtin_amdgcn_s_memtime() - tq;  // epilogue
  }
  if (prof && threadIdx.x == 0) {
    unsigned long long tk1 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 8; ++i) atomicAdd(&prof[i], pt[i]);
    atomicAdd(&prof[8], tk1 - tk0);   // cycles in kernel (this workgroup)
    atomicAdd(&prof[9], tr1 - tr0);   // 100 MHz ticks in kernel
    atomicAdd(&prof[10], 1ull);       // workgroups
  }
#undef SF_STAMP
}

// ---------------------------------------------------------------------------------------------
// "Direct" variant for 3x3x3 kernels: no LDS, no barriers, no compaction.  A wave owns 16*MB
// output rows and keeps their sums in the MFMA accumulators for the whole layer; per offset every
// lane loads its own A fragment straight from the input row the neighbour table names (zeros
// where there is none) and the filter fragments from L1/L2 (every wave reads the same 27 small
// matrices).  It multiplies the empty (row, offset) slots too, so it pays where the matrix pipe
// is not the bound (narrow layers) or where the per-phase bookkeeping of the tiled kernels costs
// more than the empty slots (64 channels).  Summation order per output: centre offset first,
// then k ascending, channel groups ascending -- fixed, no atomics.
template <int CIN, int COUT, int MB, int D1>
__global__ void __launch_bounds__(256)
sparse_conv_f32_direct_kernel(const float *__restrict__ feat, const int *__restrict__ nbr,
                              int nbr_stride, int k_first, const float *__restrict__ wpk,
                              const float *__restrict__ scale, const float *__restrict__ shift,
                              const float *__restrict__ residual, int relu, float *__restrict__ out,
                              const int *n_out_dev, int n_out_bound) {
  constexpr int K = 27;
  constexpr int CINP = CIN < 16 ? 16 : CIN;
  constexpr int T = CINP / 16, CH = CIN / 4, NB = COUT / 16, ROWS = 16 * MB;
  constexpr int D0 = 2;  // neighbour indices are loaded D0 offsets ahead of the rows they name
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, ln = lane & 15;
  const int n_out = load_n(n_out_dev, n_out_bound);
  const int row0 = (blockIdx.x * 4 + wave) * ROWS;
  if (row0 >= n_out) return;
  const f32x4 *__restrict__ wp4 = reinterpret_cast<const f32x4 *>(wpk);
  const int kf = (k_first >= 0 && k_first < K) ? k_first : -1;

  f32x4 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // every array below is indexed by compile-time constants only (the loop is fully unrolled):
  // each element is a value with a short live range, not storage
  int idx[K][MB];
  f32x4 a[K][MB][T];
  f32x4 b[K][NB][T];
#pragma unroll
  for (int i = -(D0 + D1); i < K; ++i) {
    if (i + D0 + D1 < K) {  // neighbour indices of offset i + D0 + D1 (visit order)
      const int kk = i + D0 + D1;
      const int k = kf < 0 ? kk : (kk == 0 ? kf : (kk <= kf ? kk - 1 : kk));
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        // branch-free (rows past the end re-read the last row's entry, empty slots read row
        // 0): straight-line code keeps the compiler's outstanding-load counts exact, so the
        // loads issued for later offsets really stay in flight under the MFMAs
        const int r = row0 + mb * 16 + ln;
        const int v = nbr[(size_t)k * nbr_stride + (r < n_out ? r : n_out - 1)];
        idx[kk][mb] = (r < n_out) ? v : -1;
      }
    }
    if (i + D1 >= 0 && i + D1 < K) {  // rows and filter fragments of offset i + D1
      const int kk = i + D1;
      const int k = kf < 0 ? kk : (kk == 0 ? kf : (kk <= kf ? kk - 1 : kk));
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int t = 0; t < T; ++t) b[kk][nb][t] = wp4[(((size_t)k * NB + nb) * T + t) * 64 + lane];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int ix = idx[kk][mb];  // zeroed at use, not here: the load stays in flight
          a[kk][mb][t] = *reinterpret_cast<const f32x4 *>(
              feat + (size_t)(ix >= 0 ? ix : 0) * CIN + 16 * t + ((CIN >= 16 || g < CH) ? 4 * g : 0));
        }
    }
    if (i >= 0) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const bool ok = idx[i][mb] >= 0 && (CIN >= 16 || g < CH);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) a[i][mb][t][j] = ok ? a[i][mb][t][j] : 0.0f;
      }
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][mb][t][j], b[i][nb][t][j],
                                                                 acc[mb][nb], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // epilogue: accumulator element r of lane (g, ln) is C[4g + r][ln]
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int col = nb * 16 + ln;
    const float sc = scale ? scale[col] : 1.0f, sh = scale ? shift[col] : 0.0f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + mb * 16 + 4 * g + r;
        if (row < n_out) {
          float v = acc[mb][nb][r] * sc + sh;
          if (residual) v += residual[(size_t)row * COUT + col];
          if (relu) v = v > 0.0f ? v : 0.0f;
          out[(size_t)row * COUT + col] = v;
        }
      }
  }
}

// (K, cin, cout) -> fragment order [k][nb][t][lane] float4, element j = W[k][16t+4(lane>>4)+j][16nb+(lane&15)]
__global__ void pack_filters_kernel(const float *__restrict__ w, int K, int cin, int cinp, int cout,
                                    float *__restrict__ wp) {
  const int T = cinp / 16, NB = cout / 16;
  const long long total = (long long)K * NB * T * 64 * 4;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    int j = (int)(e & 3);
    int lane = (int)((e >> 2) & 63);
    long long rest = e >> 8;
    int t = (int)(rest % T);
    rest /= T;
    int nb = (int)(rest % NB);
    int k = (int)(rest / NB);
    int c = 16 * t + 4 * (lane >> 4) + j, n = 16 * nb + (lane & 15);
    wp[e] = (c < cin) ? w[((size_t)k * cin + c) * cout + n] : 0.0f;
  }
}

__global__ void pad_channels_kernel(const float *__restrict__ src, int cs, float *__restrict__ dst,
                                    int cd, const int *n_dev, int n_bound) {
  const int n = load_n(n_dev, n_bound);
  const long long total = (long long)n * cd;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    int r = (int)(e / cd), c = (int)(e - (long long)r * cd);
    dst[e] = c < cs ? src[(size_t)r * cs + c] : 0.0f;
  }
}

unsigned long long *g_conv_prof = nullptr;  // diagnostic only, see sf_debug_conv_profile
const int *g_row_order = nullptr;            // experiment hook, see sf_debug_conv_row_order

using ConvArgs = SfConvArgs;

// Balanced tiling (32-row units handed out over the resident workgroups) pays where a layer has
// only ~2 tiles per resident workgroup (the 128-channel stage: 33k rows / 64 = 520 tiles on 256
// CUs); each call site passes the measured choice for its shape.
template <int CIN, int COUT, int TM, int AM, int WAVES, int ABUF>
int launch(const ConvArgs &a, int balance) {
  constexpr int TMP = (TM + 15) & ~15;
  constexpr size_t lds = (size_t)(TM + 1) * (COUT + 4) * 4 + (size_t)ABUF * AM * CIN * 4 +
                         (size_t)kMaxK * TMP * 5 + (size_t)kMaxK * (((TM + AM - 1) / AM) + 1) * 4 +
                         64;
  static_assert(lds <= 160 * 1024, "tile does not fit LDS");
  // grid = what is resident at once (CUs x workgroups per CU), never more than the tiles
  static SfSlotCache cache = {};
  const int slots = sf_resident_slots(cache, sparse_conv_f32_kernel<CIN, COUT, TM, AM, WAVES, ABUF>,
                                      WAVES * 64, 0, lds);
  SF_CHECK_ARG(slots > 0, "cannot query the device for the resident workgroups");
  int tiles = sf_div_up(a.n_out_bound, balance ? 32 : TM);
  int cap = balance ? slots : 4096;
  int grid = tiles < cap ? tiles : cap;
  sf_note_kernel("sparse_conv_f32_kernel<%d, %d, %d, %d, %d, %d>", CIN, COUT, TM, AM, WAVES, ABUF);
  hipLaunchKernelGGL((sparse_conv_f32_kernel<CIN, COUT, TM, AM, WAVES, ABUF>), dim3(grid),
                     dim3(WAVES * 64), 0, a.st, a.feat, a.nbr, a.nbr_stride, a.K, a.k_first, a.wpk,
                     a.scale, a.shift, a.residual, a.relu, a.out, a.n_out_dev, a.n_out_bound,
                     balance, g_conv_prof, g_row_order);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

template <int CIN, int COUT, int TM, int AM, int WAVES, int MINW = 1>
int launch_wide(const ConvArgs &a, int balance) {
  constexpr bool ok = (COUT / 16) % WAVES == 0 && (AM * (CIN / 4)) % (WAVES * 64) == 0 && CIN >= 32;
  if constexpr (!ok) {  // shape outside the wide kernel: the general kernel at its usual tile
    return launch<CIN, COUT, 80, 64, 4, 1>(a, 1);
  } else {
    using S = WideShape<CIN, COUT, TM, AM, WAVES>;
    constexpr size_t lds = (size_t)(TM + 1) * S::OST * 4 + (size_t)2 * AM * CIN * 4 +
                           (size_t)kMaxK * S::TMP * 5 + (size_t)kMaxK * 4 + (S::NMETA + 1) * 4 + 64;
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    static SfSlotCache cache = {};
    const int slots = sf_resident_slots(cache, sparse_conv_f32_wide_kernel<CIN, COUT, TM, AM, WAVES, MINW>,
                                        WAVES * 64, 0, lds);
    SF_CHECK_ARG(slots > 0, "cannot query the device for the resident workgroups");
      int tiles = sf_div_up(a.n_out_bound, balance ? 32 : TM);
    int cap = balance ? slots : 4096;
    int grid = tiles < cap ? tiles : cap;
    sf_note_kernel("sparse_conv_f32_wide_kernel<%d, %d, %d, %d, %d, %d>", CIN, COUT, TM, AM, WAVES, MINW);
    hipLaunchKernelGGL((sparse_conv_f32_wide_kernel<CIN, COUT, TM, AM, WAVES, MINW>), dim3(grid),
                       dim3(WAVES * 64), 0, a.st, a.feat, a.nbr, a.nbr_stride, a.K, a.k_first, a.wpk,
                       a.scale, a.shift, a.residual, a.relu, a.out, a.n_out_dev, a.n_out_bound,
                       balance, g_conv_prof, g_row_order);
    SF_CHECK_LAUNCH();
    return SF_OK;
  }
}

template <int CIN, int COUT, int MB, int D1>
int launch_direct(const ConvArgs &a) {
  const int rows_per_block = 16 * MB * 4;
  sf_note_kernel("sparse_conv_f32_direct_kernel<%d, %d, %d, %d>", CIN, COUT, MB, D1);
  hipLaunchKernelGGL((sparse_conv_f32_direct_kernel<CIN, COUT, MB, D1>),
                     dim3(sf_div_up(a.n_out_bound, rows_per_block)), dim3(256), 0, a.st, a.feat,
                     a.nbr, a.nbr_stride, a.k_first, a.wpk, a.scale, a.shift, a.residual, a.relu,
                     a.out, a.n_out_dev, a.n_out_bound);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// The tile of each shape is the measured best of the sweeps recorded in DESIGN.md (tile rows,
// gather chunk, waves, general / wide kernel, balanced or plain tiling).
template <int CIN, int COUT>
int launch_cfg(const ConvArgs &a) {
  // the narrow 27-offset layers whose table is too large for the narrow kernel: direct kernel
  if constexpr (CIN <= 32 && COUT <= 32 && (COUT == CIN || COUT == 2 * CIN)) {
    if (a.K == 27) return launch_direct<CIN, COUT, 1, 2>(a);
  }
  // shapes of the SparseEncoder's wide stages
  constexpr bool big = CIN >= 32 && (COUT == CIN || COUT == 2 * CIN);
  if constexpr (!big) {
    return launch<CIN, COUT, 64, 64, 4, 2>(a, 0);
  } else if constexpr (CIN >= 128) {
    return launch_wide<CIN, COUT, 96, 64, 8>(a, 1);
  } else if constexpr (COUT >= 128) {
    return launch_wide<CIN, COUT, 68, 64, 8>(a, 0);
  } else if constexpr (COUT >= 64) {  // 32-row chunks, plain 64-row tiles: 0.775 vs 0.807 ms per 4 layers
    return launch_wide<CIN, COUT, 64, 32, 4>(a, 0);
  } else {
    return launch<CIN, COUT, 80, 64, 4, 1>(a, 1);
  }
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

extern "C" int sf_conv_padded_cin(int cin) {
  if (cin < 1) return -1;
  if (cin <= 8) return 8;
  if (cin <= 16) return 16;
  if (cin <= 32) return 32;
  if (cin <= 64) return 64;
  if (cin <= 128) return 128;
  return -1;
}

extern "C" int sf_pack_filters_f32(const float *w_kio, int K, int cin, int cout, float *w_packed,
                                   void *stream) {
  int cp = sf_conv_padded_cin(cin);
  SF_CHECK_ARG(w_kio && w_packed, "filters pointer is NULL");
  SF_CHECK_ARG(K >= 1 && K <= kMaxK, "kernel volume %d not in [1,%d]", K, kMaxK);
  SF_CHECK_ARG(cp > 0, "unsupported cin=%d (1..128)", cin);
  SF_CHECK_ARG(cout == 16 || cout == 32 || cout == 64 || cout == 128, "unsupported cout=%d", cout);
  int cinp = cp < 16 ? 16 : cp;
  long long total = (long long)K * cinp * cout;
  hipLaunchKernelGGL(pack_filters_kernel, dim3(sf_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, w_kio, K, cin, cinp, cout, w_packed);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_pad_channels_f32(const float *src, int c_src, float *dst, int c_dst,
                                   const int *n_dev, int n_bound, void *stream) {
  SF_CHECK_ARG(c_src >= 1 && c_dst >= c_src && n_bound >= 0, "bad channel counts / n_bound");
  if (n_bound == 0) return SF_OK;
  SF_CHECK_ARG(src && dst, "src / dst is NULL");
  hipLaunchKernelGGL(pad_channels_kernel, dim3(sf_grid((long long)n_bound * c_dst, 256)), dim3(256),
                     0, (hipStream_t)stream, src, c_src, dst, c_dst, n_dev, n_bound);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_sparse_conv_f32(const float *features_in, int n_in_bound, int cin_padded, const int *nbr,
                                  int nbr_stride, int K, int k_first, const float *w_packed,
                                  int cout, const float *scale, const float *shift,
                                  const float *residual, int relu, float *out,
                                  const int *n_out_dev, int n_out_bound, void *stream) {
  hipStream_t st = (hipStream_t)stream;
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
  SF_CHECK_ARG(n_in_bound >= 1, "n_in_bound=%d: the input feature table must hold a row", n_in_bound);
  // the narrow kernel gathers through a raw buffer bounded by the table size and marks a missing
  // neighbour with offset 2^31: tables of 2 GiB and more stay on the general kernels (64-bit
  // addresses)
  const long long feat_bytes = (long long)n_in_bound * cin_padded * 4;
  const bool small_table = feat_bytes < (1ll << 31);
  ConvArgs a{features_in, small_table ? (unsigned)feat_bytes : 0u, nbr, nbr_stride, K, k_first,
             w_packed, scale, shift, residual, relu, out, n_out_dev, n_out_bound, st};
  if (small_table) {  // the narrow 3x3x3 layers have a kernel of their own
    const int rc = sf_narrow_conv_f32(cin_padded, cout, a);
    if (rc <= 0) return rc;
  }
  switch (cin_padded) {
    case 8: return dispatch_cout<8>(cout, a);
    case 16: return dispatch_cout<16>(cout, a);
    case 32: return dispatch_cout<32>(cout, a);
    case 64: return dispatch_cout<64>(cout, a);
    case 128: return dispatch_cout<128>(cout, a);
  }
  sf_set_error("unsupported cin_padded=%d (8, 16, 32, 64 or 128; see sf_conv_padded_cin)",
               cin_padded);
  return SF_ERR_BAD_ARG;
}

// Diagnostic hook (not part of the documented ABI surface used by callers): when set to a device
// buffer of 16 uint64, every conv launch accumulates wave-0 cycle stamps per region into it.
extern "C" __attribute__((visibility("default"))) void sf_debug_conv_profile(void *dev_buf16) {
  g_conv_prof = (unsigned long long *)dev_buf16;
}
// Experiment hook: a row order (sf_row_order_by_mask) for the tiles of the wide kernel; NULL = off.
extern "C" __attribute__((visibility("default"))) void sf_debug_conv_row_order(const void *order) {
  g_row_order = (const int *)order;
}

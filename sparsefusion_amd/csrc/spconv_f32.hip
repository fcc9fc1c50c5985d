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
#include "sf_common.h"

namespace {

constexpr int kTM = 64;       // output rows per tile (= one wavefront for the ballot compaction)
constexpr int kThreads = 256; // 4 waves
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

template <int CIN, int COUT>
__global__ void __launch_bounds__(kThreads)
sparse_conv_f32_kernel(const float *__restrict__ feat, const int *__restrict__ nbr, int nbr_stride,
                       int K, int k_first, const float *__restrict__ wpk,
                       const float *__restrict__ scale, const float *__restrict__ shift,
                       const float *__restrict__ residual, int relu, float *__restrict__ out,
                       const int *n_out_dev, int n_out_bound) {
  constexpr int CINP = CIN < 16 ? 16 : CIN;  // packed-weight K extent
  constexpr int T = CINP / 16;               // 16-channel groups = float4 B fragments per N block
  constexpr int CH = CIN / 4;
  constexpr int NB = COUT / 16;
  constexpr int CG = NB < 4 ? NB : 4;        // wave groups along output columns
  constexpr int NBW = NB / CG;               // N blocks per wave
  constexpr int RG = 4 / CG;                 // wave groups along compact rows
  constexpr int OST = COUT + 4;              // padded output-tile row stride (floats)
  constexpr int GI = (kTM * CH + kThreads - 1) / kThreads;  // gather items per thread

  __shared__ __attribute__((aligned(16))) float s_out[kTM * OST];
  __shared__ __attribute__((aligned(16))) float s_a[2][kTM * CIN];
  __shared__ int s_src[kMaxK][kTM];
  __shared__ unsigned char s_dst[kMaxK][kTM];
  __shared__ int s_cnt[kMaxK];
  __shared__ int s_order[kMaxK + 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = wave % CG, rg = wave / CG;
  const int g = lane >> 4, ln = lane & 15;
  const int n_out = load_n(n_out_dev, n_out_bound);
  const f32x4 *__restrict__ wp4 = reinterpret_cast<const f32x4 *>(wpk);

  for (int tile = blockIdx.x; tile * kTM < n_out; tile += gridDim.x) {
    const int row0 = tile * kTM;
    // ---- setup: compact the tile's pairs per offset, clear the output tile --------------
    for (int k = wave; k < K; k += 4) {
      int o = row0 + lane;
      int v = (o < n_out) ? nbr[(size_t)k * nbr_stride + o] : -1;
      unsigned long long m = __ballot(v >= 0);
      int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                          __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      if (v >= 0) {
        s_src[k][pos] = v;
        s_dst[k][pos] = (unsigned char)lane;
      }
      if (lane == 0) s_cnt[k] = __popcll(m);
    }
    for (int i = tid; i < kTM * OST; i += kThreads) s_out[i] = 0.0f;
    __syncthreads();
    if (tid == 0) {  // visit order: k_first (SubM centre) then ascending, empty offsets skipped
      int no = 0;
      if (k_first >= 0 && k_first < K && s_cnt[k_first] > 0) s_order[no++] = k_first;
      for (int k = 0; k < K; ++k)
        if (k != k_first && s_cnt[k] > 0) s_order[no++] = k;
      s_order[kMaxK] = no;
    }
    __syncthreads();
    const int n_order = s_order[kMaxK];

    f32x4 greg[GI];
    // gather of the first offset
    if (n_order > 0) {
      const int k = s_order[0], cnt = s_cnt[k];
#pragma unroll
      for (int m = 0; m < GI; ++m) {
        int idx = tid + m * kThreads, p = idx / CH, c = idx % CH;
        if (idx < kTM * CH && p < cnt)
          greg[m] = *reinterpret_cast<const f32x4 *>(feat + (size_t)s_src[k][p] * CIN + 4 * c);
      }
#pragma unroll
      for (int m = 0; m < GI; ++m) {
        int idx = tid + m * kThreads, p = idx / CH, c = idx % CH;
        if (idx < kTM * CH && p < cnt)
          *reinterpret_cast<f32x4 *>(&s_a[0][p * CIN + 4 * ATile<CIN>::pos(p, c)]) = greg[m];
      }
    }
    __syncthreads();

    for (int kk = 0; kk < n_order; ++kk) {
      const int k = s_order[kk], cnt = s_cnt[k];
      const int buf = kk & 1;
      // (1) issue the next offset's gather loads (land in registers while we compute)
      int cnt_next = 0, k_next = 0;
      if (kk + 1 < n_order) {
        k_next = s_order[kk + 1];
        cnt_next = s_cnt[k_next];
#pragma unroll
        for (int m = 0; m < GI; ++m) {
          int idx = tid + m * kThreads, p = idx / CH, c = idx % CH;
          if (idx < kTM * CH && p < cnt_next)
            greg[m] =
                *reinterpret_cast<const f32x4 *>(feat + (size_t)s_src[k_next][p] * CIN + 4 * c);
        }
      }
      // (2) this offset: B fragments from the packed filters (L2 resident), MFMA, add to tile
      const int nsb = (cnt + 15) >> 4;
      if (rg < nsb) {
        f32x4 bfrag[NBW][T];
#pragma unroll
        for (int i = 0; i < NBW; ++i)
#pragma unroll
          for (int t = 0; t < T; ++t)
            bfrag[i][t] = wp4[(((size_t)k * NB + (cg * NBW + i)) * T + t) * 64 + lane];
        for (int sb = rg; sb < nsb; sb += RG) {
          f32x4 acc[NBW];
#pragma unroll
          for (int i = 0; i < NBW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
          const int p = sb * 16 + ln;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            f32x4 a4 = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (CIN >= 16 || g < CH)
              a4 = *reinterpret_cast<const f32x4 *>(
                  &s_a[buf][p * CIN + 4 * ATile<CIN>::pos(p, 4 * t + g)]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int i = 0; i < NBW; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], bfrag[i][t][j], acc[i], 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int q = sb * 16 + g * 4 + r;
            if (q < cnt) {
              int dst = s_dst[k][q];
#pragma unroll
              for (int i = 0; i < NBW; ++i)
                s_out[dst * OST + (cg * NBW + i) * 16 + ln] += acc[i][r];
            }
          }
        }
      }
      // (3) park the prefetched rows in the other LDS buffer
      if (kk + 1 < n_order) {
#pragma unroll
        for (int m = 0; m < GI; ++m) {
          int idx = tid + m * kThreads, p = idx / CH, c = idx % CH;
          if (idx < kTM * CH && p < cnt_next)
            *reinterpret_cast<f32x4 *>(&s_a[buf ^ 1][p * CIN + 4 * ATile<CIN>::pos(p, c)]) = greg[m];
        }
      }
      __syncthreads();
    }

    // ---- epilogue: folded BN, residual, ReLU; one coalesced write per output row ---------
    constexpr int OC = COUT / 4;
    for (int idx = tid; idx < kTM * OC; idx += kThreads) {
      int r = idx / OC, c = idx % OC, row = row0 + r;
      if (row >= n_out) continue;
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

template <int CIN, int COUT>
int launch(const float *feat, const int *nbr, int nbr_stride, int K, int k_first, const float *wpk,
           const float *scale, const float *shift, const float *residual, int relu, float *out,
           const int *n_out_dev, int n_out_bound, hipStream_t st) {
  int tiles = sf_div_up(n_out_bound, kTM);
  int grid = tiles < 2048 ? tiles : 2048;
  hipLaunchKernelGGL((sparse_conv_f32_kernel<CIN, COUT>), dim3(grid), dim3(kThreads), 0, st, feat,
                     nbr, nbr_stride, K, k_first, wpk, scale, shift, residual, relu, out, n_out_dev,
                     n_out_bound);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

template <int CIN>
int dispatch_cout(int cout, const float *feat, const int *nbr, int nbr_stride, int K, int k_first,
                  const float *wpk, const float *scale, const float *shift, const float *residual,
                  int relu, float *out, const int *n_out_dev, int n_out_bound, hipStream_t st) {
#define SF_CASE(CO)                                                                             \
  case CO:                                                                                      \
    return launch<CIN, CO>(feat, nbr, nbr_stride, K, k_first, wpk, scale, shift, residual, relu, \
                           out, n_out_dev, n_out_bound, st);
  switch (cout) {
    SF_CASE(16)
    SF_CASE(32)
    SF_CASE(64)
    SF_CASE(128)
  }
#undef SF_CASE
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

extern "C" int sf_sparse_conv_f32(const float *features_in, int cin_padded, const int *nbr,
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
#define SF_CIN(CI)                                                                              \
  case CI:                                                                                      \
    return dispatch_cout<CI>(cout, features_in, nbr, nbr_stride, K, k_first, w_packed, scale,   \
                             shift, residual, relu, out, n_out_dev, n_out_bound, st);
  switch (cin_padded) {
    SF_CIN(8)
    SF_CIN(16)
    SF_CIN(32)
    SF_CIN(64)
    SF_CIN(128)
  }
#undef SF_CIN
  sf_set_error("unsupported cin_padded=%d (8, 16, 32, 64 or 128; see sf_conv_padded_cin)",
               cin_padded);
  return SF_ERR_BAD_ARG;
}

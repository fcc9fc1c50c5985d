// Dense 2-D convolution + folded BatchNorm + ReLU for the BEV backbone and neck, gfx950
// (SECOND, mmdet3d/models/backbones/second.py:36-86, and SECONDFPN,
// mmdet3d/models/necks/second_fpn.py:42-92 -- nn.Conv2d / nn.ConvTranspose2d(k = s) /
// nn.BatchNorm2d / nn.ReLU chains on the (B, 256, 180, 180) map the sparse encoder returns).
//
// Implicit GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains), NCHW in
// and out so that the module boundary needs no layout change:
//   * a workgroup owns TH output rows x 32 output columns x 64 output channels; wave r owns row r;
//   * the product is taken transposed, out^T = W^T * patch: the filters are MFMA operand A, the
//     pixels operand B, so a lane ends with 4 output channels of ONE pixel column and 16 lanes
//     cover 16 consecutive x of a channel plane -> 64-byte NCHW stores with no LDS transpose;
//   * input channels go through LDS 8 at a time (32 for 1x1), double buffered, one barrier per
//     stage, every global load of the next stage issued branch-free before the current stage's
//     MFMAs (addresses clamped, values masked; the fill plan of a thread is computed once): the
//     input patch (TH*S+KS-S rows x 32*S+KS-S columns, zero padded at the borders, read from HBM/L2
//     once per stage and reused by all KS*KS taps and all 64 output channels), and the stage's
//     filters, which the host packed in fragment order so that the fill is a straight 16-byte copy
//     and one ds_read_b128 per (tap, 4 channels) yields the A operands of four 16-channel groups;
//   * the channel stride of the patch image is padded to 16 (mod 32) dwords (stride 1) or an odd
//     number (stride 2) so the ds_read_b32 of operand B (16 pixels x 2 channels per LDS cycle) is
//     conflict free;
//   * per (tap, 4 channels) a wave issues 1 ds_read_b128 + 2 ds_read_b32 for 8 MFMAs (256 matrix
//     cycles) on 8 independent accumulators;
//   * transposed convolutions with kernel = stride (`up`) are `up*up` independent 1x1 convolutions
//     whose results interleave in the output: they run as ONE 1x1 launch over up*up*Cout packed
//     output channels ordered (dy, co, dx); the epilogue sends channel (dy, co, dx) of pixel
//     (y, x) to (co, up*y + dy, up*x + dx), dx pairs / quads of a lane as one 8 / 16-byte store;
//   * the epilogue applies y = acc*scale[co] + shift[co] (eval BatchNorm folded by the caller),
//     ReLU, and writes into channels [out_coff, out_coff + Cout) of a buffer with out_ctot
//     channels, so SECONDFPN's torch.cat (second_fpn.py:88-89) costs nothing.
// Bound: fp32 MFMA (AI of a 128->128 3x3 layer at 180x180 = 9.5 GFLOP / 34 MB).
#include <stdlib.h>

#include "sf_common.h"

namespace {

constexpr int TW = 32;  // output columns per workgroup row
constexpr int TN = 64;  // output channels per workgroup
// input channels per LDS stage: 8 for 3x3 / 2x2 (72 / 32 MFMAs per wave and barrier), 32 for 1x1
constexpr int stage_channels(int ks) { return ks == 1 ? 32 : 8; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Conv2dArgs {
  int B, Cin, H, W, Ho, Wo, pad;
  int cout_packed;  // up*up*cout: channels of the implicit GEMM
  int cout;         // channels per tap = length of scale / shift
  int up, out_ctot, out_coff, relu;
  int nct, nxt, nyt, nchunks, n_wg;
};

template <int KS, int S, int TH, int NG = 4>
struct Geo {
  static constexpr int CK = stage_channels(KS);
  static constexpr int CKG = CK / 4;
  static constexpr int TAPS = KS * KS;
  static constexpr int PH = (TH - 1) * S + KS;
  static constexpr int PW = (TW - 1) * S + KS;
  static constexpr int CS = (S == 1) ? ((PH * PW + 15) / 32 * 32 + 16) : ((PH * PW) | 1);
  static constexpr int W4 = TAPS * CKG * 64;  // float4 per filter stage of a 64-channel tile
  static constexpr int WV = W4;                // per-lane filter pieces of this workgroup's stage
  static constexpr int W_FLOATS = WV * NG;     // (NG floats each)
  static constexpr int P_ELEMS = CK * PH * PW;
  static constexpr int DUMMY = CK * CS;          // patch slot that absorbs out-of-range fill lanes
  static constexpr int STAGE = W_FLOATS + CK * CS + 4;
  static constexpr int THREADS = TH * 64;
  static constexpr int NW4 = (WV + THREADS - 1) / THREADS;
  static constexpr int NP = (P_ELEMS + THREADS - 1) / THREADS;
  static constexpr size_t LDS_BYTES = 2u * STAGE * sizeof(float);
  static_assert(NP <= 32, "okmask holds one bit per patch element of a thread");
};

template <int KS, int S, int TH, int NG>
__global__ void __launch_bounds__(TH * 64)
conv2d_f32_kernel(const float *__restrict__ in, const f32x4 *__restrict__ wp,
                  const float *__restrict__ scale, const float *__restrict__ shift,
                  float *__restrict__ out, const Conv2dArgs a) {
  using G = Geo<KS, S, TH, NG>;
  typedef float wvec __attribute__((ext_vector_type(NG)));  // a lane's filter piece
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wr = tid >> 6;

  // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs, so give each XCD
  // a contiguous run of tiles (the output-channel tiles of one patch are neighbours in the run
  // and share the patch through that XCD's L2).
  const int per = (a.n_wg + 7) >> 3;
  int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (t >= a.n_wg) return;
  // channel tile: nct tiles of 16*NG channels; the packed filters are laid out per 64 channels,
  // a 32-channel workgroup reads one half of every lane's float4
  const int ctn = t % a.nct;
  t /= a.nct;
  const int ct = ctn * NG / 4, half = (NG == 4) ? 0 : (ctn & 1);
  const int xt = t % a.nxt;
  t /= a.nxt;
  const int yt = t % a.nyt, b = t / a.nyt;
  const int oy0 = yt * TH, ox0 = xt * TW;
  const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
  const size_t plane = (size_t)a.H * a.W;
  const float *__restrict__ inb = in + (size_t)b * a.Cin * plane;
  const f32x4 *__restrict__ wct = wp + (size_t)ct * a.nchunks * G::W4;

  // Fill plan of this thread, fixed for the whole layer: element e of the patch image ->
  // offset inside the channel-chunk of the input (or -1 outside the map) and LDS slot.
  int goff[G::NP], loff[G::NP];
#pragma unroll
  for (int i = 0; i < G::NP; ++i) {
    const int e = tid + i * G::THREADS;
    const int c = e / (G::PH * G::PW), rem = e - c * (G::PH * G::PW);
    const int r = rem / G::PW, j = rem - r * G::PW;
    const int iy = iy0 + r, ix = ix0 + j;
    const bool in_map = e < G::P_ELEMS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    goff[i] = in_map ? c * (int)plane + iy * a.W + ix : -1;
    loff[i] = e < G::P_ELEMS ? c * G::CS + rem : G::DUMMY;
  }

  wvec wreg[G::NW4];
  float preg[G::NP];
  unsigned okmask = 0;  // bit i: preg[i] is a real input element (else the slot is zero padding)

  // All loads are unconditional (clamped address; the value is masked only when it is written to
  // LDS, after the current stage's MFMAs): no branch and no early wait, so every load of the next
  // stage stays in flight under the matrix work.
  auto load_stage = [&](int ch) {
#pragma unroll
    for (int i = 0; i < G::NW4; ++i) {
      int idx = tid + i * G::THREADS;
      if (G::WV % G::THREADS != 0) idx = idx < G::WV ? idx : G::WV - 1;
      wreg[i] = *reinterpret_cast<const wvec *>(
          reinterpret_cast<const float *>(wct + (size_t)ch * G::W4 + idx) + half * NG);
    }
    const int c0 = ch * G::CK, cbase = c0 * (int)plane;
    okmask = 0;
#pragma unroll
    for (int i = 0; i < G::NP; ++i) {
      const int e = tid + i * G::THREADS;
      const bool ok = goff[i] >= 0 && (a.Cin % G::CK == 0 || c0 + e / (G::PH * G::PW) < a.Cin);
      preg[i] = inb[ok ? cbase + goff[i] : 0];
      okmask |= (unsigned)ok << i;
    }
  };
  auto store_stage = [&](int buf) {
    float *st = lds + buf * G::STAGE;
#pragma unroll
    for (int i = 0; i < G::NW4; ++i) {
      const int idx = tid + i * G::THREADS;
      if (G::WV % G::THREADS == 0 || idx < G::WV) reinterpret_cast<wvec *>(st)[idx] = wreg[i];
    }
#pragma unroll
    for (int i = 0; i < G::NP; ++i)
      st[G::W_FLOATS + loff[i]] = (okmask >> i) & 1u ? preg[i] : 0.0f;
  };

  f32x4 acc[NG][2];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int p = 0; p < 2; ++p) acc[g][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool row_live = oy0 + wr < a.Ho;
  const int pb = (lane >> 4) * G::CS + wr * S * G::PW + (lane & 15) * S;

  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int ch = 0; ch < a.nchunks; ++ch) {
    const bool more = ch + 1 < a.nchunks;
    if (more) load_stage(ch + 1);
    if (row_live) {
      const float *st = lds + (ch & 1) * G::STAGE;
      const wvec *wl = reinterpret_cast<const wvec *>(st) + lane;
      const float *pl = st + G::W_FLOATS + pb;
      // operands of step s+1 (one tap x 4 channels) are read from LDS while step s multiplies
      constexpr int NS = G::TAPS * G::CKG;
      wvec wa[2];
      float b0[2], b1[2];
      auto fetch = [&](int s, int slot) {
        const int tap = s / G::CKG, cg = s - tap * G::CKG, ky = tap / KS, kx = tap - ky * KS;
        wa[slot] = wl[s * 64];
        b0[slot] = pl[cg * 4 * G::CS + ky * G::PW + kx];
        b1[slot] = pl[cg * 4 * G::CS + ky * G::PW + kx + 16 * S];
      };
      if constexpr (KS == 1) {
        // 8 short steps per stage: the compiler's own schedule is better here (pinned order
        // measured 65 vs 55 us on the 1x1 neck layer)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          fetch(s, 0);
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0][g], b0[0], acc[g][0], 0, 0, 0);
            acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0][g], b1[0], acc[g][1], 0, 0, 0);
          }
        }
      } else {
        fetch(0, 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          if (s + 1 < NS) fetch(s + 1, (s + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            acc[g][0] =
                __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s & 1][g], b0[s & 1], acc[g][0], 0, 0, 0);
            acc[g][1] =
                __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s & 1][g], b1[s & 1], acc[g][1], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (more) store_stage((ch + 1) & 1);
    __syncthreads();
  }

  // epilogue: lane holds packed channels cp0 + i (cp0 = n0 + g*16 + 4*(lane>>4), i = 0..3) of pixel
  // columns ox0 + p*16 + (lane&15).  Packed channel order is (dy, co, dx): for up = 2 / 4 the
  // lane's consecutive channels are consecutive x of one output row -> 8 / 16-byte stores.
  if (!row_live) return;
  const int oy = oy0 + wr;
  const int up = a.up, HO = a.Ho * up, WO = a.Wo * up;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int cp0 = ctn * 16 * NG + g * 16 + 4 * (lane >> 4);
    if (up == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = cp0 + i;
        if (co >= a.cout) continue;
        const float sc = scale ? scale[co] : 1.0f, sh = shift ? shift[co] : 0.0f;
        float *__restrict__ orow = out + (((size_t)b * a.out_ctot + a.out_coff + co) * HO + oy) * WO;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int ox = ox0 + p * 16 + (lane & 15);
          if (ox < a.Wo) {
            float v = acc[g][p][i] * sc + sh;
            if (a.relu) v = fmaxf(v, 0.0f);
            orow[ox] = v;
          }
        }
      }
    } else if (up == 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int cq = (cp0 >> 1) + h;  // (dy, co)
        if (cq >= 2 * a.cout) continue;
        const int dy = cq / a.cout, co = cq - dy * a.cout;
        const float sc = scale ? scale[co] : 1.0f, sh = shift ? shift[co] : 0.0f;
        float *__restrict__ orow =
            out + (((size_t)b * a.out_ctot + a.out_coff + co) * HO + (size_t)oy * 2 + dy) * WO;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int ox = ox0 + p * 16 + (lane & 15);
          if (ox < a.Wo) {
            float2 v = make_float2(acc[g][p][2 * h] * sc + sh, acc[g][p][2 * h + 1] * sc + sh);
            if (a.relu) v.x = fmaxf(v.x, 0.0f), v.y = fmaxf(v.y, 0.0f);
            *reinterpret_cast<float2 *>(orow + (size_t)ox * 2) = v;
          }
        }
      }
    } else if (up == 4) {
      const int cq = cp0 >> 2;  // (dy, co)
      if (cq >= 4 * a.cout) continue;
      const int dy = cq / a.cout, co = cq - dy * a.cout;
      const float sc = scale ? scale[co] : 1.0f, sh = shift ? shift[co] : 0.0f;
      float *__restrict__ orow =
          out + (((size_t)b * a.out_ctot + a.out_coff + co) * HO + (size_t)oy * 4 + dy) * WO;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int ox = ox0 + p * 16 + (lane & 15);
        if (ox < a.Wo) {
          float4 v = make_float4(acc[g][p][0] * sc + sh, acc[g][p][1] * sc + sh,
                                 acc[g][p][2] * sc + sh, acc[g][p][3] * sc + sh);
          if (a.relu)
            v.x = fmaxf(v.x, 0.0f), v.y = fmaxf(v.y, 0.0f), v.z = fmaxf(v.z, 0.0f),
            v.w = fmaxf(v.w, 0.0f);
          *reinterpret_cast<float4 *>(orow + (size_t)ox * 4) = v;
        }
      }
    } else {  // any other upsampling factor: scalar stores
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cp = cp0 + i;
        if (cp >= a.cout_packed) continue;
        const int dx = cp % up, cq = cp / up, dy = cq / a.cout, co = cq - dy * a.cout;
        const float sc = scale ? scale[co] : 1.0f, sh = shift ? shift[co] : 0.0f;
        float *__restrict__ orow =
            out + (((size_t)b * a.out_ctot + a.out_coff + co) * HO + (size_t)oy * up + dy) * WO + dx;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int ox = ox0 + p * 16 + (lane & 15);
          if (ox < a.Wo) {
            float v = acc[g][p][i] * sc + sh;
            if (a.relu) v = fmaxf(v, 0.0f);
            orow[(size_t)ox * up] = v;
          }
        }
      }
    }
  }
}

// (Cout', Cin, KS, KS) -> [cout tile][stage][tap][4-channel group][lane][4 channel groups of 16];
// for an upsampling layer the caller's rows are already in packed order (dy, co, dx)
__global__ void conv2d_pack_kernel(const float *__restrict__ w, int coutp, int cin, int taps,
                                   int nct, int nchunks, int ckg, f32x4 *__restrict__ wp) {
  const long long total = (long long)nct * nchunks * taps * ckg * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    long long r = i >> 6;
    const int cg = (int)(r % ckg);
    r /= ckg;
    const int tap = (int)(r % taps);
    r /= taps;
    const int ch = (int)(r % nchunks), ct = (int)(r / nchunks);
    const int ci = (ch * ckg + cg) * 4 + (lane >> 4);
    f32x4 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = ct * TN + g * 16 + (lane & 15);
      v[g] = (co < coutp && ci < cin) ? w[((size_t)co * cin + ci) * taps + tap] : 0.0f;
    }
    wp[i] = v;
  }
}

template <int KS, int S, int TH, int NG>
int launch(const float *in, const float *wp, const float *scale, const float *shift, float *out,
           Conv2dArgs a, hipStream_t st) {
  using G = Geo<KS, S, TH, NG>;
  auto kern = conv2d_f32_kernel<KS, S, TH, NG>;
  static bool attr_set[kSfMaxDevices] = {};  // per instantiation and device: the attribute is sticky
  if (G::LDS_BYTES > 48 * 1024) {
    int dev = 0;
    SF_CHECK_HIP(hipGetDevice(&dev));
    SF_CHECK_ARG(dev >= 0 && dev < kSfMaxDevices, "device ordinal %d out of range", dev);
    if (!attr_set[dev]) {
      SF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
      attr_set[dev] = true;
    }
  }
  a.nyt = sf_div_up(a.Ho, TH);
  a.nct = sf_div_up(a.cout_packed, 16 * NG);
  a.n_wg = a.B * a.nyt * a.nxt * a.nct;
  const int grid = ((a.n_wg + 7) / 8) * 8;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(G::THREADS), G::LDS_BYTES, st, in,
                     reinterpret_cast<const f32x4 *>(wp), scale, shift, out, a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

template <int KS, int S>
int dispatch_th(int th, int ng, const float *in, const float *wp, const float *scale,
                const float *shift, float *out, const Conv2dArgs &a, hipStream_t st) {
  if (ng == 2) {
    if (th == 6) return launch<KS, S, 6, 2>(in, wp, scale, shift, out, a, st);
    return launch<KS, S, 4, 2>(in, wp, scale, shift, out, a, st);
  }
  if (th == 6) return launch<KS, S, 6, 4>(in, wp, scale, shift, out, a, st);
  return launch<KS, S, 4, 4>(in, wp, scale, shift, out, a, st);
}

// Channels per workgroup: 64, or 32 where the 64-channel grid is about two workgroups per CU (one
// 180x180 map at batch 1: 540 workgroups on 256 CUs; 1080 half-width ones measured 4 % faster:
// 131 vs 137 us for 128->128).  Smaller grids (a 90x90 map: 276) and the 1x1 / stride-2 layers
// measured SLOWER with 32 channels (169 vs 161 us, 92 vs 80 us): they keep 64.
int pick_groups(const Conv2dArgs &a, int th) {
  const long long wgs = (long long)a.B * sf_div_up(a.Ho, th) * a.nxt * sf_div_up(a.cout_packed, 64);
  return (wgs >= 400 && wgs < 4 * 256 && a.cout_packed % 32 == 0) ? 2 : 4;
}

// Rows (= waves) per workgroup.  4 puts one wave of every workgroup on each SIMD and measured
// best whenever the grid is several workgroups per CU (B = 8: 107 TFLOP/s against 82-86 with 3 or
// 6 rows, whose waves load the four SIMDs unevenly; 2 rows: 66).  6 rows are taken only where
// they save a whole round of workgroups on the 256 CUs (a 90x90 map at batch 1: 180 workgroups
// instead of 276).
int pick_rows(const Conv2dArgs &a) {
  auto cost = [&](int th) {
    const long long wgs = (long long)a.B * sf_div_up(a.Ho, th) * a.nxt * a.nct;
    return ((wgs + 255) / 256) * th;
  };
  return 10 * cost(6) <= 8 * cost(4) ? 6 : 4;
}

}  // namespace

extern "C" size_t sf_conv2d_packed_bytes(int cout, int cin, int ksize, int up) {
  if (cout < 1 || cin < 1 || ksize < 1 || up < 1) return 0;
  const int ck = stage_channels(ksize);
  const size_t nct = (size_t)sf_div_up((long long)cout * up * up, TN), nch = sf_div_up(cin, ck);
  return nct * nch * ksize * ksize * (ck / 4) * 64 * sizeof(f32x4);
}

extern "C" int sf_conv2d_pack_filters_f32(const float *weight, int cout, int cin, int ksize, int up,
                                          float *packed, void *stream) {
  SF_CHECK_ARG(weight && packed, "weight / packed is NULL");
  SF_CHECK_ARG(cout >= 1 && cin >= 1 && up >= 1, "bad filter shape");
  SF_CHECK_ARG(ksize >= 1 && ksize <= 3, "kernel size %d not supported (1, 2, 3)", ksize);
  SF_CHECK_ARG(up == 1 || ksize == 1, "an upsampling layer is packed as up*up 1x1 filters");
  const int ck = stage_channels(ksize);
  const int coutp = cout * up * up, nct = sf_div_up(coutp, TN), nch = sf_div_up(cin, ck);
  const long long total = (long long)nct * nch * ksize * ksize * (ck / 4) * 64;
  hipLaunchKernelGGL(conv2d_pack_kernel, dim3(sf_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, weight, coutp, cin, ksize * ksize, nct, nch, ck / 4,
                     reinterpret_cast<f32x4 *>(packed));
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_conv2d_f32(const float *in, const float *packed, const float *scale,
                             const float *shift, float *out, int batch, int cin, int h, int w,
                             int cout, int ksize, int stride, int pad, int up, int out_channels,
                             int out_channel_offset, int relu, void *stream) {
  SF_CHECK_ARG(in && packed && out, "in / packed / out is NULL");
  SF_CHECK_ARG(batch >= 1 && cin >= 1 && cout >= 1 && h >= 1 && w >= 1, "bad conv2d shape");
  SF_CHECK_ARG(up >= 1 && (up == 1 || (ksize == 1 && stride == 1 && pad == 0)),
               "an upsampling layer runs as a 1x1 / stride 1 / pad 0 launch with up > 1");
  SF_CHECK_ARG(pad >= 0 && pad < ksize, "bad padding %d", pad);
  SF_CHECK_ARG(out_channel_offset >= 0 && out_channel_offset + cout <= out_channels,
               "channels [%d, %d) do not fit an output of %d channels", out_channel_offset,
               out_channel_offset + cout, out_channels);
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (w + 2 * pad - ksize) / stride + 1;
  SF_CHECK_ARG(h + 2 * pad >= ksize && w + 2 * pad >= ksize, "input smaller than the kernel");
  Conv2dArgs a;
  a.B = batch, a.Cin = cin, a.H = h, a.W = w, a.Ho = ho, a.Wo = wo, a.pad = pad;
  a.cout_packed = cout * up * up, a.cout = cout, a.up = up;
  a.out_ctot = out_channels, a.out_coff = out_channel_offset, a.relu = relu;
  a.nct = sf_div_up(a.cout_packed, TN), a.nxt = sf_div_up(wo, TW), a.nyt = 0;
  a.nchunks = sf_div_up(cin, stage_channels(ksize)), a.n_wg = 0;
  SF_CHECK_ARG((long long)cin * h * w < (1ll << 31), "one sample of the input must stay below 2^31 elements");
  SF_CHECK_ARG((long long)batch * sf_div_up(ho, 2) * a.nxt * a.nct < (1ll << 30), "grid too large");
  hipStream_t st = (hipStream_t)stream;
  const int th = pick_rows(a), ng = pick_groups(a, th);
  if (ksize == 3 && stride == 1) return dispatch_th<3, 1>(th, ng, in, packed, scale, shift, out, a, st);
  if (ksize == 3 && stride == 2) return dispatch_th<3, 2>(th, ng, in, packed, scale, shift, out, a, st);
  if (ksize == 1 && stride == 1) return dispatch_th<1, 1>(th, ng, in, packed, scale, shift, out, a, st);
  if (ksize == 2 && stride == 2) return dispatch_th<2, 2>(th, ng, in, packed, scale, shift, out, a, st);
  sf_set_error("conv2d kernel %d / stride %d not supported (3/1, 3/2, 1/1, 2/2)", ksize, stride);
  return SF_ERR_BAD_ARG;
}

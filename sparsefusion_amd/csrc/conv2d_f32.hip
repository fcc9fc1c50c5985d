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
//   * input channels go through LDS 8 at a time, double buffered, one barrier per stage: the
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
//     output channels, the epilogue sends channel (tap, co) of pixel (y, x) to
//     (co, up*y + tap/up, up*x + tap%up);
//   * the epilogue applies y = acc*scale[co] + shift[co] (eval BatchNorm folded by the caller),
//     ReLU, and writes into channels [out_coff, out_coff + Cout) of a buffer with out_ctot
//     channels, so SECONDFPN's torch.cat (second_fpn.py:88-89) costs nothing.
// Bound: fp32 MFMA (AI of a 128->128 3x3 layer at 180x180 = 9.5 GFLOP / 34 MB).
#include <stdlib.h>

#include "sf_common.h"

namespace {

constexpr int TW = 32;  // output columns per workgroup row
constexpr int TN = 64;  // output channels per workgroup
constexpr int CK = 8;   // input channels per LDS stage
constexpr int CKG = CK / 4;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Conv2dArgs {
  int B, Cin, H, W, Ho, Wo, pad;
  int cout_packed;  // up*up*cout: channels of the implicit GEMM
  int cout;         // channels per tap = length of scale / shift
  int up, out_ctot, out_coff, relu;
  int nct, nxt, nyt, nchunks, n_wg;
};

template <int KS, int S, int TH>
struct Geo {
  static constexpr int TAPS = KS * KS;
  static constexpr int PH = (TH - 1) * S + KS;
  static constexpr int PW = (TW - 1) * S + KS;
  static constexpr int CS = (S == 1) ? ((PH * PW + 15) / 32 * 32 + 16) : ((PH * PW) | 1);
  static constexpr int W4 = TAPS * CKG * 64;  // float4 per filter stage
  static constexpr int W_FLOATS = W4 * 4;
  static constexpr int P_ELEMS = CK * PH * PW;
  static constexpr int STAGE = W_FLOATS + CK * CS;
  static constexpr int THREADS = TH * 64;
  static constexpr int NW4 = (W4 + THREADS - 1) / THREADS;
  static constexpr int NP = (P_ELEMS + THREADS - 1) / THREADS;
  static constexpr size_t LDS_BYTES = 2u * STAGE * sizeof(float);
};

template <int KS, int S, int TH>
__global__ void __launch_bounds__(TH * 64)
conv2d_f32_kernel(const float *__restrict__ in, const f32x4 *__restrict__ wp,
                  const float *__restrict__ scale, const float *__restrict__ shift,
                  float *__restrict__ out, const Conv2dArgs a) {
  using G = Geo<KS, S, TH>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wr = tid >> 6;

  // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs, so give each XCD
  // a contiguous run of tiles (the output-channel tiles of one patch are neighbours in the run
  // and share the patch through that XCD's L2).
  const int per = (a.n_wg + 7) >> 3;
  int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (t >= a.n_wg) return;
  const int ct = t % a.nct;
  t /= a.nct;
  const int xt = t % a.nxt;
  t /= a.nxt;
  const int yt = t % a.nyt, b = t / a.nyt;
  const int oy0 = yt * TH, ox0 = xt * TW;
  const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
  const size_t plane = (size_t)a.H * a.W;
  const float *__restrict__ inb = in + (size_t)b * a.Cin * plane;
  const f32x4 *__restrict__ wct = wp + (size_t)ct * a.nchunks * G::W4;

  f32x4 wreg[G::NW4];
  float preg[G::NP];

  auto load_stage = [&](int ch) {
#pragma unroll
    for (int i = 0; i < G::NW4; ++i) {
      const int idx = tid + i * G::THREADS;
      if (G::W4 % G::THREADS == 0 || idx < G::W4) wreg[i] = wct[(size_t)ch * G::W4 + idx];
    }
    const int c0 = ch * CK;
#pragma unroll
    for (int i = 0; i < G::NP; ++i) {
      const int e = tid + i * G::THREADS;
      const int c = e / (G::PH * G::PW), rem = e - c * (G::PH * G::PW);
      const int r = rem / G::PW, j = rem - r * G::PW;
      const int iy = iy0 + r, ix = ix0 + j;
      const bool ok = e < G::P_ELEMS && c0 + c < a.Cin && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      preg[i] = ok ? inb[(size_t)(c0 + c) * plane + (size_t)iy * a.W + ix] : 0.0f;
    }
  };
  auto store_stage = [&](int buf) {
    float *st = lds + buf * G::STAGE;
#pragma unroll
    for (int i = 0; i < G::NW4; ++i) {
      const int idx = tid + i * G::THREADS;
      if (G::W4 % G::THREADS == 0 || idx < G::W4) reinterpret_cast<f32x4 *>(st)[idx] = wreg[i];
    }
#pragma unroll
    for (int i = 0; i < G::NP; ++i) {
      const int e = tid + i * G::THREADS;
      const int c = e / (G::PH * G::PW), rem = e - c * (G::PH * G::PW);
      if (e < G::P_ELEMS) st[G::W_FLOATS + c * G::CS + rem] = preg[i];
    }
  };

  f32x4 acc[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g)
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
      const f32x4 *wl = reinterpret_cast<const f32x4 *>(st) + lane;
      const float *pl = st + G::W_FLOATS + pb;
#pragma unroll
      for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
          for (int cg = 0; cg < CKG; ++cg) {
            const f32x4 wa = wl[((ky * KS + kx) * CKG + cg) * 64];
            const float b0 = pl[cg * 4 * G::CS + ky * G::PW + kx];
            const float b1 = pl[cg * 4 * G::CS + ky * G::PW + kx + 16 * S];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[g], b0, acc[g][0], 0, 0, 0);
              acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[g], b1, acc[g][1], 0, 0, 0);
            }
          }
    }
    if (more) store_stage((ch + 1) & 1);
    __syncthreads();
  }

  // epilogue: lane holds channels n0 + g*16 + 4*(lane>>4) + i of pixel columns ox0 + p*16 + (lane&15)
  if (!row_live) return;
  const int oy = oy0 + wr;
  const int up = a.up, HO = a.Ho * up, WO = a.Wo * up;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cp = ct * TN + g * 16 + 4 * (lane >> 4) + i;
      if (cp >= a.cout_packed) continue;
      const int tap = cp / a.cout, co = cp - tap * a.cout;
      const float sc = scale ? scale[co] : 1.0f, sh = shift ? shift[co] : 0.0f;
      const int dy = tap / up, dx = tap - dy * up;
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

// (Cout', Cin, KS, KS) -> [cout tile][stage][tap][4-channel group][lane][4 channel groups of 16]
__global__ void conv2d_pack_kernel(const float *__restrict__ w, int coutp, int cin, int taps,
                                   int nct, int nchunks, f32x4 *__restrict__ wp) {
  const long long total = (long long)nct * nchunks * taps * CKG * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    long long r = i >> 6;
    const int cg = (int)(r % CKG);
    r /= CKG;
    const int tap = (int)(r % taps);
    r /= taps;
    const int ch = (int)(r % nchunks), ct = (int)(r / nchunks);
    const int ci = ch * CK + cg * 4 + (lane >> 4);
    f32x4 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = ct * TN + g * 16 + (lane & 15);
      v[g] = (co < coutp && ci < cin) ? w[((size_t)co * cin + ci) * taps + tap] : 0.0f;
    }
    wp[i] = v;
  }
}

template <int KS, int S, int TH>
int launch(const float *in, const float *wp, const float *scale, const float *shift, float *out,
           Conv2dArgs a, hipStream_t st) {
  using G = Geo<KS, S, TH>;
  auto kern = conv2d_f32_kernel<KS, S, TH>;
  static bool attr_set = false;  // per instantiation; the attribute is sticky per function
  if (G::LDS_BYTES > 48 * 1024 && !attr_set) {
    SF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
    attr_set = true;
  }
  a.nyt = sf_div_up(a.Ho, TH);
  a.n_wg = a.B * a.nyt * a.nxt * a.nct;
  const int grid = ((a.n_wg + 7) / 8) * 8;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(G::THREADS), G::LDS_BYTES, st, in,
                     reinterpret_cast<const f32x4 *>(wp), scale, shift, out, a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

template <int KS, int S>
int dispatch_th(int th, const float *in, const float *wp, const float *scale, const float *shift,
                float *out, const Conv2dArgs &a, hipStream_t st) {
  switch (th) {
    case 2: return launch<KS, S, 2>(in, wp, scale, shift, out, a, st);
    case 3: return launch<KS, S, 3>(in, wp, scale, shift, out, a, st);
    case 4: return launch<KS, S, 4>(in, wp, scale, shift, out, a, st);
    default: return launch<KS, S, 6>(in, wp, scale, shift, out, a, st);
  }
}

// Rows per workgroup: the candidate whose busiest CU finishes first (workgroups are dealt evenly
// over 256 CUs; a workgroup's time is proportional to its rows), larger tiles on ties because the
// filters are re-read once per workgroup.
int pick_rows(const Conv2dArgs &a, int stride) {
  if (const char *e = getenv("SF_CONV2D_TH")) {
    const int v = atoi(e);
    if (v == 2 || v == 3 || v == 4 || v == 6) return v;
  }
  static const int cand[4] = {6, 4, 3, 2};
  int best = 2;
  long long best_cost = -1;
  for (int i = 0; i < 4; ++i) {
    const int th = cand[i];
    if (stride == 2 && th == 6) continue;  // patch image would not leave room for 2 workgroups per CU
    const long long wgs = (long long)a.B * sf_div_up(a.Ho, th) * a.nxt * a.nct;
    const long long cost = ((wgs + 255) / 256) * th;
    if (best_cost < 0 || cost < best_cost) best_cost = cost, best = th;
  }
  return best;
}

}  // namespace

extern "C" size_t sf_conv2d_packed_bytes(int cout, int cin, int ksize, int up) {
  if (cout < 1 || cin < 1 || ksize < 1 || up < 1) return 0;
  const size_t nct = (size_t)sf_div_up((long long)cout * up * up, TN), nch = sf_div_up(cin, CK);
  return nct * nch * ksize * ksize * CKG * 64 * sizeof(f32x4);
}

extern "C" int sf_conv2d_pack_filters_f32(const float *weight, int cout, int cin, int ksize, int up,
                                          float *packed, void *stream) {
  SF_CHECK_ARG(weight && packed, "weight / packed is NULL");
  SF_CHECK_ARG(cout >= 1 && cin >= 1 && up >= 1, "bad filter shape");
  SF_CHECK_ARG(ksize >= 1 && ksize <= 3, "kernel size %d not supported (1, 2, 3)", ksize);
  SF_CHECK_ARG(up == 1 || ksize == 1, "an upsampling layer is packed as up*up 1x1 filters");
  const int coutp = cout * up * up, nct = sf_div_up(coutp, TN), nch = sf_div_up(cin, CK);
  const long long total = (long long)nct * nch * ksize * ksize * CKG * 64;
  hipLaunchKernelGGL(conv2d_pack_kernel, dim3(sf_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, weight, coutp, cin, ksize * ksize, nct, nch,
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
  a.nchunks = sf_div_up(cin, CK), a.n_wg = 0;
  SF_CHECK_ARG((long long)batch * sf_div_up(ho, 2) * a.nxt * a.nct < (1ll << 30), "grid too large");
  hipStream_t st = (hipStream_t)stream;
  const int th = pick_rows(a, stride);
  if (ksize == 3 && stride == 1) return dispatch_th<3, 1>(th, in, packed, scale, shift, out, a, st);
  if (ksize == 3 && stride == 2) return dispatch_th<3, 2>(th, in, packed, scale, shift, out, a, st);
  if (ksize == 1 && stride == 1) return dispatch_th<1, 1>(th, in, packed, scale, shift, out, a, st);
  if (ksize == 2 && stride == 2) return dispatch_th<2, 2>(th, in, packed, scale, shift, out, a, st);
  sf_set_error("conv2d kernel %d / stride %d not supported (3/1, 3/2, 1/1, 2/2)", ksize, stride);
  return SF_ERR_BAD_ARG;
}

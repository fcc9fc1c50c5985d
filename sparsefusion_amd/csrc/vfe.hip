// Voxel feature encoder layer (HardVFE / VFELayer, PillarFeatureNet / PFNLayer) for gfx950.
//
// PillarFeatureNet.forward (mmdet3d/models/voxel_encoders/pillar_encoder.py:91-150) + PFNLayer
// (voxel_encoders/utils.py:106-180) are the same computation with a 2-channel voxel-centre offset
// and, in legacy mode, raw x / y overwritten by it: decorate bits 16 and 32.
//
// Reference: HardVFE.forward (mmdet3d/models/voxel_encoders/voxel_encoder.py:383-450) decorates
// the (M, P, C) voxel tensor (cluster-centre offsets :407-413, voxel-centre offsets :416-428,
// distance :430-432), zeroes the padded slots (:438-441) and runs VFELayer.forward
// (mmdet3d/models/voxel_encoders/utils.py:61-96): Linear(bias=False) -> BatchNorm1d -> ReLU ->
// max over the P slots [-> concat with the point-wise features].  That is 6-8 torch kernels and
// three (M, P, C) temporaries per layer; here one kernel per layer reads the voxel tensor once
// and writes the layer output once (HBM bound: M*P*C*4 read + M*Cout*4 write).
//
// BatchNorm is the eval-mode fold y = x*scale + shift.  Note the reference takes the max over ALL
// P slots: a padded slot contributes relu(shift) (its masked input is zero), and so does this.
#include "sf_common.h"

namespace {

__device__ __forceinline__ int load_n(const int *n_dev, int n_bound) {
  int n = n_dev ? *n_dev : n_bound;
  return n < n_bound ? n : n_bound;
}

struct VfeDeco {
  int cluster, center, dist, mask;
  int center_xy;  // PillarFeatureNet: the voxel-centre offset has x and y only (pillar_encoder.py:114-131)
  int legacy;     // PillarFeatureNet(legacy=True): the offsets also overwrite raw x, y (:124-130)
  float vs[3];   // voxel size x, y, z
  float off[3];  // vs/2 + range_min per axis (voxel_encoder.py:363-365)
};

// TPV threads work on one voxel (thread = output channel); a 256-thread block holds 256/TPV voxels.
template <int TPV>
__global__ void __launch_bounds__(256)
vfe_layer_kernel(const float *__restrict__ in, int cin_raw, int P, const int *__restrict__ num_points,
                 const int *__restrict__ coors4, const int *m_dev, int m_bound, VfeDeco d,
                 const float *__restrict__ weight, int cin, int cout,
                 const float *__restrict__ scale, const float *__restrict__ shift, int mode,
                 float *__restrict__ out) {
  constexpr int VPB = 256 / TPV;
  extern __shared__ float smem[];
  float *wt = smem;                      // [cin][cout]  (transposed: lanes read consecutive c)
  float *xs = wt + cin * cout;           // [VPB][P][cin]
  float *mean = xs + VPB * P * cin;      // [VPB][4]
  const int m = load_n(m_dev, m_bound);
  const int tid = threadIdx.x, vl = tid / TPV, c = tid % TPV;
  for (int i = tid; i < cin * cout; i += 256) {
    const int f = i / cout, cc = i - f * cout;
    wt[i] = weight[(size_t)cc * cin + f];  // nn.Linear.weight is (cout, cin)
  }
  const int groups = (m + VPB - 1) / VPB;
  for (int gi = blockIdx.x; gi < groups; gi += gridDim.x) {
    const int v = gi * VPB + vl;
    const bool live = v < m;
    float *x = xs + (size_t)vl * P * cin;
    const int np = (live && num_points) ? num_points[v] : 0;  // NULL when nothing needs it
    __syncthreads();  // previous group's xs reads are done (and wt is in place)
    if (live) {
      for (int i = c; i < P * cin_raw; i += TPV) {
        const int s = i / cin_raw, f = i - s * cin_raw;
        x[s * cin + f] = in[((size_t)v * P + s) * cin_raw + f];
      }
    }
    __syncthreads();
    if (live && d.cluster && c < 3) {  // mean of xyz over ALL slots / num_points (:408-410)
      float sum = 0.0f;
      for (int s = 0; s < P; ++s) sum += x[s * cin + c];
      mean[vl * 4 + c] = sum / (float)np;
    }
    __syncthreads();
    if (live) {
      int4 cr = make_int4(0, 0, 0, 0);
      if (d.center) cr = reinterpret_cast<const int4 *>(coors4)[v];
      for (int s = c; s < P; s += TPV) {
        float *xr = x + s * cin;
        const bool deco = d.cluster || d.center || d.dist;  // then cin_raw >= 3 (host check)
        const float px = deco ? xr[0] : 0.0f, py = deco ? xr[1] : 0.0f, pz = deco ? xr[2] : 0.0f;
        int f = cin_raw;
        if (d.cluster) {
          xr[f++] = px - mean[vl * 4 + 0];
          xr[f++] = py - mean[vl * 4 + 1];
          xr[f++] = pz - mean[vl * 4 + 2];
        }
        float dx = px, dy = py;  // what the distance decoration sees as x, y
        if (d.center) {  // coors are (b, z, y, x)
          const float cx = px - ((float)cr.w * d.vs[0] + d.off[0]);
          const float cy = py - ((float)cr.z * d.vs[1] + d.off[1]);
          xr[f++] = cx;
          xr[f++] = cy;
          if (!d.center_xy) xr[f++] = pz - ((float)cr.y * d.vs[2] + d.off[2]);
          if (d.legacy) {
            // the reference's f_center is a VIEW of features[:, :, :2] modified in place
            // (pillar_encoder.py:124-130): raw x, y become the offsets too, after f_cluster was
            // taken from the original values and before the distance is
            xr[0] = cx, xr[1] = cy;
            dx = cx, dy = cy;
          }
        }
        if (d.dist) xr[f++] = sqrtf(dx * dx + dy * dy + pz * pz);
        if (d.mask && s >= np)
          for (int q = 0; q < cin; ++q) xr[q] = 0.0f;
      }
    }
    __syncthreads();
    if (live && c < cout) {
      const float sc = scale[c], sh = shift[c];
      float best = 0.0f;  // ReLU output is >= 0, so 0 is the identity of the max
      for (int s = 0; s < P; ++s) {
        const float *xr = x + s * cin;
        float acc = 0.0f;
        for (int f = 0; f < cin; ++f) acc = fmaf(xr[f], wt[f * cout + c], acc);
        float y = acc * sc + sh;
        y = y > 0.0f ? y : 0.0f;
        best = y > best ? y : best;
        if (mode == 1) out[((size_t)v * P + s) * cout + c] = y;
        if (mode == 2) out[((size_t)v * P + s) * 2 * cout + c] = y;
      }
      if (mode == 0) out[(size_t)v * cout + c] = best;
      if (mode == 2)
        for (int s = 0; s < P; ++s) out[((size_t)v * P + s) * 2 * cout + cout + c] = best;
    }
  }
}

}  // namespace

extern "C" int sf_vfe_layer(const float *inputs, int cin_raw, int max_points, const int *num_points,
                            const int *coors4, const int *m_dev, int m_bound, int decorate,
                            const float voxel_size[3], const float coors_range[6],
                            const float *weight, int cout, const float *scale, const float *shift,
                            int mode, float *out, void *stream) {
  SF_CHECK_ARG(m_bound >= 0 && cin_raw >= 1 && max_points >= 1, "bad m_bound/cin/max_points");
  SF_CHECK_ARG(cout >= 1 && cout <= 256, "cout=%d not in [1,256]", cout);
  SF_CHECK_ARG(mode >= 0 && mode <= 2, "mode must be 0 (max), 1 (point-wise) or 2 (concat)");
  SF_CHECK_ARG((decorate & ~63) == 0, "unknown decorate bits");
  SF_CHECK_ARG(!(decorate & 48) || (decorate & 2), "decorate bits 16 / 32 qualify bit 2");
  if (m_bound == 0) return SF_OK;
  SF_CHECK_ARG(inputs && weight && scale && shift && out, "NULL tensor pointer");
  SF_CHECK_ARG(!(decorate & 9) || num_points, "num_points is NULL");
  SF_CHECK_ARG(!(decorate & 7) || cin_raw >= 3, "decorations need xyz in the first 3 channels");
  SF_CHECK_ARG(!(decorate & 2) || (coors4 && voxel_size && coors_range),
               "voxel-centre offsets need coors / voxel_size / range");
  VfeDeco d;
  d.cluster = decorate & 1;
  d.center = (decorate >> 1) & 1;
  d.dist = (decorate >> 2) & 1;
  d.mask = (decorate >> 3) & 1;
  d.center_xy = (decorate >> 4) & 1;
  d.legacy = (decorate >> 5) & 1;
  for (int i = 0; i < 3; ++i) {
    d.vs[i] = voxel_size ? voxel_size[i] : 0.0f;
    d.off[i] = (voxel_size && coors_range) ? voxel_size[i] / 2 + coors_range[i] : 0.0f;
  }
  const int cin = cin_raw + 3 * d.cluster + (d.center_xy ? 2 : 3) * d.center + d.dist;
  const int tpv = cout <= 32 ? 32 : (cout <= 64 ? 64 : (cout <= 128 ? 128 : 256));
  const int vpb = 256 / tpv;
  const size_t lds = ((size_t)cin * cout + (size_t)vpb * max_points * cin + vpb * 4) * sizeof(float);
  SF_CHECK_ARG(lds <= 64 * 1024, "VFE layer of %d x %d channels x %d points needs %zu bytes of LDS "
               "(> 64 KiB)", cin, cout, max_points, lds);
  int grid = sf_div_up(m_bound, vpb);
  if (grid > 2048) grid = 2048;
  hipStream_t st = (hipStream_t)stream;
#define SF_VFE_LAUNCH(T)                                                                          \
  hipLaunchKernelGGL((vfe_layer_kernel<T>), dim3(grid), dim3(256), lds, st, inputs, cin_raw,      \
                     max_points, num_points, coors4, m_dev, m_bound, d, weight, cin, cout, scale, \
                     shift, mode, out)
  switch (tpv) {
    case 32: SF_VFE_LAUNCH(32); break;
    case 64: SF_VFE_LAUNCH(64); break;
    case 128: SF_VFE_LAUNCH(128); break;
    default: SF_VFE_LAUNCH(256); break;
  }
#undef SF_VFE_LAUNCH
  SF_CHECK_LAUNCH();
  return SF_OK;
}

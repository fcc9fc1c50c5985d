/*
 * sf_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's CPU algorithms for the LiDAR hot
 * path (voxelize -> per-voxel mean -> spconv-1.x rulebook + gather/GEMM/
 * scatter-add -> BN/ReLU -> dense BEV).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (sparsefusion_amd/) never does and fails loudly without its HIP library.
 *
 * Every function cites the reference file:line (relative to the upstream
 * checkout) whose arithmetic it follows.  Nothing here is copied; the loops
 * are re-derived from the reference's behaviour.
 *
 * Pinning: sfo_hard_voxelize / sfo_dynamic_voxelize are checked against the
 * reference's own known-answer test (tests/test_models/test_voxel_encoder/
 * test_voxel_generator.py:15-19) and against the reference's CPU voxelizer
 * compiled from its own sources into oracle/_ref (see oracle/build_ref.py).
 * The rulebook / convolution functions follow spconv 1.x as vendored in the
 * reference (mmdet3d/ops/spconv/include/spconv/geometry.h, spconv_ops.h);
 * that code needs <cuda_runtime_api.h>, which this image lacks, so it is not
 * built here and those functions are pinned by a dense torch conv3d
 * equivalence instead (tests/test_oracle_conv_dense.py): the reference's own
 * tests hold no value fixtures for them ("parity unpinned" by upstream).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define SFO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* voxelization                                                         */
/* ------------------------------------------------------------------ */

/* grid_size[i] = round((hi - lo) / vs) evaluated in fp32
 * (voxelization_cpu.cpp:119-122, :155-158). */
SFO_API void sfo_grid_size(const float vs[3], const float range[6], int grid[3]) {
  for (int i = 0; i < 3; ++i) {
    float q = (range[3 + i] - range[i]) / vs[i];
    grid[i] = (int)roundf(q);
  }
}

/* One point -> (z,y,x) or fail.  voxelization_cpu.cpp:20-37.
 * The reference converts floor() of a float to int with a plain C cast; for
 * NaN/Inf/huge values that cast is undefined behaviour and on x86-64 yields
 * INT_MIN, which the (c < 0) test then rejects.  The oracle states that
 * outcome explicitly instead of relying on UB. */
static inline int sfo_point_coor(const float *p, const float vs[3], const float range[6],
                                 const int grid[3], int coor_zyx[3]) {
  for (int j = 0; j < 3; ++j) {
    float v = floorf((p[j] - range[j]) / vs[j]);
    if (!(v >= 0.0f && v < (float)grid[j])) return 0; /* NaN fails too */
    coor_zyx[2 - j] = (int)v;
  }
  return 1;
}

/* dynamic_voxelize_kernel, voxelization_cpu.cpp:7-41 (all three -1 on fail). */
SFO_API void sfo_dynamic_voxelize(const float *pts, int n, int f, const float vs[3],
                                  const float range[6], int *coors) {
  int grid[3];
  sfo_grid_size(vs, range, grid);
  for (int i = 0; i < n; ++i) {
    int c[3];
    if (sfo_point_coor(pts + (size_t)i * f, vs, range, grid, c)) {
      coors[3 * i + 0] = c[0];
      coors[3 * i + 1] = c[1];
      coors[3 * i + 2] = c[2];
    } else {
      coors[3 * i + 0] = coors[3 * i + 1] = coors[3 * i + 2] = -1;
    }
  }
}

/* hard_voxelize_kernel + hard_voxelize_cpu, voxelization_cpu.cpp:43-142.
 * voxels (max_voxels,max_points,f), coors (max_voxels,3), npv (max_voxels)
 * are caller-zeroed as in voxelize.py:46-50.  Returns voxel_num, or -1 when
 * the dense lookup grid cannot be allocated. */
SFO_API int sfo_hard_voxelize(const float *pts, int n, int f, const float vs[3],
                              const float range[6], int max_points, int max_voxels,
                              float *voxels, int *coors, int *npv) {
  int grid[3];
  sfo_grid_size(vs, range, grid);
  size_t vol = (size_t)grid[0] * grid[1] * grid[2];
  /* dense coor_to_voxelidx[z][y][x] filled with -1 (cpp:127-128) */
  int *lut = (int *)malloc(vol * sizeof(int));
  if (!lut) return -1;
  memset(lut, 0xFF, vol * sizeof(int));
  int voxel_num = 0;
  for (int i = 0; i < n; ++i) {
    int c[3];
    if (!sfo_point_coor(pts + (size_t)i * f, vs, range, grid, c)) continue;
    size_t cell = ((size_t)c[0] * grid[1] + c[1]) * grid[0] + c[2];
    int vid = lut[cell];
    if (vid == -1) {
      vid = voxel_num;
      if (max_voxels != -1 && voxel_num >= max_voxels) break; /* cpp:78 */
      voxel_num += 1;
      lut[cell] = vid;
      coors[3 * vid + 0] = c[0];
      coors[3 * vid + 1] = c[1];
      coors[3 * vid + 2] = c[2];
    }
    int num = npv[vid];
    if (max_points == -1 || num < max_points) {
      memcpy(voxels + ((size_t)vid * max_points + num) * f, pts + (size_t)i * f,
             sizeof(float) * f);
      npv[vid] = num + 1;
    }
  }
  free(lut);
  return voxel_num;
}

/* HardSimpleVFE.forward, voxel_encoder.py:42-44:
 * features[:, :, :nf].sum(dim=1) / num_points  (sum runs over all max_points
 * slots, zero padded ones included; slot order ascending). */
SFO_API void sfo_vfe_mean(const float *voxels, const int *npv, int m, int max_points, int f,
                          int nf, float *out) {
  for (int v = 0; v < m; ++v) {
    for (int k = 0; k < nf; ++k) {
      float s = 0.0f;
      for (int p = 0; p < max_points; ++p) s += voxels[((size_t)v * max_points + p) * f + k];
      out[(size_t)v * nf + k] = s / (float)npv[v];
    }
  }
}

/* extract_voxel_heights, detectors/sparsefusion.py:73-112 ("next" row f1):
 * per BEV cell (y//osf, x//osf) min / max over voxels of the per-voxel
 * min / max point z, where zero z values are treated as padding (sentinels
 * +100 / -200). */
SFO_API void sfo_voxel_heights(const float *voxels, const int *coors4, int m, int max_points,
                               int f, int batch, int ny, int nx, int osf, float *min_h,
                               float *max_h) {
  size_t cells = (size_t)batch * ny * nx;
  for (size_t i = 0; i < cells; ++i) {
    min_h[i] = 100.0f;
    max_h[i] = -200.0f;
  }
  for (int v = 0; v < m; ++v) {
    float mn = 100.0f, mx = -200.0f; /* empty reduce keeps sentinel (z==0 -> 100 / -200) */
    for (int p = 0; p < max_points; ++p) {
      float z = voxels[((size_t)v * max_points + p) * f + 2];
      float zmin = (z == 0.0f) ? 100.0f : z;
      float zmax = (z == 0.0f) ? -200.0f : z;
      /* voxels_==100 after the first substitution also maps to -200 (py:85) */
      if (zmin == 100.0f) zmax = -200.0f;
      if (zmin < mn) mn = zmin;
      if (zmax > mx) mx = zmax;
    }
    const int *c = coors4 + 4 * (size_t)v;
    size_t cell = ((size_t)c[0] * ny + c[2] / osf) * nx + c[3] / osf;
    if (mn < min_h[cell]) min_h[cell] = mn;
    if (mx > max_h[cell]) max_h[cell] = mx;
  }
}

/* ------------------------------------------------------------------ */
/* spconv 1.x rulebook                                                  */
/* ------------------------------------------------------------------ */

/* getValidOutPos<Index,3>, include/spconv/geometry.h:24-85.  out holds up to
 * kernel-volume entries of (o0,o1,o2,offset).  C integer division truncates
 * toward zero exactly as in the reference. */
static int sfo_valid_out_pos(const int *in_pos, const int *ks, const int *st, const int *pad,
                             const int *dil, const int *oshape, int *out) {
  int lowers[3], uppers[3], counter[3], csize[3];
  int npts = 1, cnt = 0;
  for (int i = 0; i < 3; ++i) {
    lowers[i] = (in_pos[i] - (ks[i] - 1) * dil[i] - 1 + st[i] + pad[i]) / st[i];
    uppers[i] = (in_pos[i] + pad[i]) / st[i];
  }
  for (int i = 0; i < 3; ++i) {
    csize[i] = (uppers[i] - lowers[i]) / dil[i] + 1;
    npts *= csize[i];
    counter[i] = 0;
  }
  for (int i = 0; i < npts; ++i) {
    int valid = 1, m = 1, offset = 0;
    for (int j = 2; j >= 0; --j) {
      int val = uppers[j] - counter[j] * dil[j];
      out[cnt * 4 + j] = val;
      if (val < 0 || val > oshape[j] - 1) valid = 0;
      offset += m * (in_pos[j] - val * st[j] + pad[j]) / dil[j];
      m *= ks[j];
    }
    out[cnt * 4 + 3] = offset;
    if (valid) ++cnt;
    counter[2] += 1;
    for (int c = 2; c >= 0; --c) {
      if (counter[c] == csize[c] && c > 0) {
        counter[c - 1] += 1;
        counter[c] = 0;
      }
    }
  }
  return cnt;
}

/* rowArrayIdx<Index,3>, include/tensorview/tensorview.h:453-464 */
static inline int64_t sfo_lin(const int *p, const int *shape) {
  return ((int64_t)p[0] * shape[1] + p[1]) * shape[2] + p[2];
}

/* getIndicePairsSubM, geometry.h:247-297 with the driver's subM overrides
 * (stride 1, padding k/2: spconv_ops.h:76-79).  indices (n,4)=(b,z,y,x).
 * pairs is (K,2,n) pre-filled with -1, num is (K) zeroed (spconv_ops.h:55-59);
 * the dense grid (batch*volume ints, -1) is allocated here (spconv_ops.h:60-62).
 * Returns n, or -1 on allocation failure. */
SFO_API int sfo_indice_pairs_subm(const int *indices, int n, int batch, const int *shape,
                                  const int *ks, const int *dil, int *pairs, int *num) {
  int st[3] = {1, 1, 1}, pad[3] = {ks[0] / 2, ks[1] / 2, ks[2] / 2};
  int K = ks[0] * ks[1] * ks[2];
  int64_t vol = (int64_t)shape[0] * shape[1] * shape[2];
  int *grid = (int *)malloc((size_t)(vol * batch) * sizeof(int));
  if (!grid) return -1;
  memset(grid, 0xFF, (size_t)(vol * batch) * sizeof(int));
  int *vp = (int *)malloc((size_t)K * 4 * sizeof(int));
  for (int i = 0; i < K * 2 * n; ++i) pairs[i] = -1;
  memset(num, 0, sizeof(int) * K);
  for (int j = 0; j < n; ++j) /* geometry.h:275-280, last writer wins */
    grid[sfo_lin(indices + 4 * (size_t)j + 1, shape) + vol * indices[4 * (size_t)j]] = j;
  for (int j = 0; j < n; ++j) {
    int nv = sfo_valid_out_pos(indices + 4 * (size_t)j + 1, ks, st, pad, dil, shape, vp);
    for (int i = 0; i < nv; ++i) {
      const int *pp = vp + 4 * i;
      int off = pp[3];
      int64_t idx = sfo_lin(pp, shape) + vol * indices[4 * (size_t)j];
      if (grid[idx] > -1) {
        pairs[((size_t)off * 2 + 0) * n + num[off]] = j;
        pairs[((size_t)off * 2 + 1) * n + num[off]] = grid[idx];
        num[off] += 1;
      }
    }
  }
  free(vp);
  free(grid);
  return n;
}

/* getIndicePairsConv, geometry.h:144-194.  out_indices has room for n*K rows
 * (spconv_ops.h:106-108); returns numActOut (outputs in first-touch order). */
SFO_API int sfo_indice_pairs_conv(const int *indices, int n, int batch, const int *oshape,
                                  const int *ks, const int *st, const int *pad, const int *dil,
                                  int *out_indices, int *pairs, int *num) {
  int K = ks[0] * ks[1] * ks[2];
  int64_t vol = (int64_t)oshape[0] * oshape[1] * oshape[2];
  int *grid = (int *)malloc((size_t)(vol * batch) * sizeof(int));
  if (!grid) return -1;
  memset(grid, 0xFF, (size_t)(vol * batch) * sizeof(int));
  int *vp = (int *)malloc((size_t)K * 4 * sizeof(int));
  for (int64_t i = 0; i < (int64_t)K * 2 * n; ++i) pairs[i] = -1;
  memset(num, 0, sizeof(int) * K);
  int nact = 0;
  for (int j = 0; j < n; ++j) {
    int b = indices[4 * (size_t)j];
    int nv = sfo_valid_out_pos(indices + 4 * (size_t)j + 1, ks, st, pad, dil, oshape, vp);
    for (int i = 0; i < nv; ++i) {
      const int *pp = vp + 4 * i;
      int off = pp[3];
      int64_t idx = sfo_lin(pp, oshape) + vol * b;
      if (grid[idx] == -1) {
        out_indices[4 * (size_t)nact + 0] = b;
        out_indices[4 * (size_t)nact + 1] = pp[0];
        out_indices[4 * (size_t)nact + 2] = pp[1];
        out_indices[4 * (size_t)nact + 3] = pp[2];
        grid[idx] = nact++;
      }
      pairs[((size_t)off * 2 + 0) * n + num[off]] = j;
      pairs[((size_t)off * 2 + 1) * n + num[off]] = grid[idx];
      num[off] += 1;
    }
  }
  free(vp);
  free(grid);
  return nact;
}

/* ------------------------------------------------------------------ */
/* spconv 1.x convolution                                               */
/* ------------------------------------------------------------------ */

/* obuf[i,:] = ibuf[i,:] * W (cin x cout), fp32, c ascending (stands in for
 * torch::mm_out, spconv_ops.h:302,338).  Rows are independent, so the OpenMP
 * split changes no value. */
static void sfo_gemm(const float *a, int rows, int cin, const float *w, int cout, float *o) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < rows; ++i) {
    float *orow = o + (size_t)i * cout;
    const float *arow = a + (size_t)i * cin;
    for (int q = 0; q < cout; ++q) orow[q] = 0.0f;
    for (int c = 0; c < cin; ++c) {
      float av = arow[c];
      const float *wrow = w + (size_t)c * cout;
      for (int q = 0; q < cout; ++q) orow[q] += av * wrow[q];
    }
  }
}

/* indiceConv<float>, spconv_ops.h:260-361 (forward, inverse=0) with the CPU
 * gather / scatter-add functors of src/reordering.cc:20-50.
 * filters (K,cin,cout) [= (kD,kH,kW,cin,cout) viewed, spconv_ops.h:299],
 * pairs (K,2,L), num (K).  out (n_out,cout) is overwritten.
 * Returns 0, or -1 on allocation failure. */
/* Wall time spent in the three stages of sfo_indice_conv since the last reset (bench.py's
 * cpu_baseline breakdown: gather / GEMM / scatter-add). */
static double sfo_t_stage[3];
static double sfo_now(void) {
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return 0.0;
#endif
}
SFO_API void sfo_conv_stage_times(double out3[3], int reset) {
  for (int i = 0; i < 3; ++i) {
    out3[i] = sfo_t_stage[i];
    if (reset) sfo_t_stage[i] = 0.0;
  }
}

SFO_API int sfo_indice_conv(const float *feat, int n_in, int cin, const float *filt, int cout,
                            const int *pairs, const int *num, int K, int L, int n_out,
                            int subm, float *out) {
  (void)n_in;
  int kmax = 0, nmax = 0;
  for (int k = 0; k < K; ++k) /* std::max_element -> first maximum (:272-277) */
    if (num[k] > nmax) {
      nmax = num[k];
      kmax = k;
    }
  memset(out, 0, sizeof(float) * (size_t)n_out * cout);
  if (nmax <= 0) return 0;
  float *ibuf = (float *)malloc(sizeof(float) * (size_t)nmax * cin);
  float *obuf = (float *)malloc(sizeof(float) * (size_t)nmax * cout);
  if (!ibuf || !obuf) {
    free(ibuf);
    free(obuf);
    return -1;
  }
  double t0 = sfo_now();
  if (subm) /* centre offset without gather/scatter (:300-303) */
    sfo_gemm(feat, n_out, cin, filt + (size_t)kmax * cin * cout, cout, out);
  sfo_t_stage[1] += sfo_now() - t0;
  for (int k = 0; k < K; ++k) {
    int nhot = num[k];
    if (nhot <= 0 || (subm && k == kmax)) continue;
    const int *pin = pairs + ((size_t)k * 2 + 0) * L;
    const int *pout = pairs + ((size_t)k * 2 + 1) * L;
    t0 = sfo_now();
    for (int i = 0; i < nhot; ++i) /* reordering.cc:26-30 */
      memcpy(ibuf + (size_t)i * cin, feat + (size_t)pin[i] * cin, sizeof(float) * cin);
    double t1 = sfo_now();
    sfo_gemm(ibuf, nhot, cin, filt + (size_t)k * cin * cout, cout, obuf);
    double t2 = sfo_now();
    for (int i = 0; i < nhot; ++i) { /* reordering.cc:42-48 */
      float *o = out + (size_t)pout[i] * cout;
      const float *b = obuf + (size_t)i * cout;
      for (int q = 0; q < cout; ++q) o[q] += b[q];
    }
    sfo_t_stage[0] += t1 - t0;
    sfo_t_stage[1] += t2 - t1;
    sfo_t_stage[2] += sfo_now() - t2;
  }
  free(ibuf);
  free(obuf);
  return 0;
}

/* Eval-mode BatchNorm1d on (n,c) + optional residual add + optional ReLU:
 * sparse_block.py:116-139 / make_sparse_convmodule :198-201 semantics,
 * y = (x-mean)/sqrt(var+eps)*gamma+beta evaluated as torch does
 * (x*scale + shift is NOT used here so the oracle stays the literal formula). */
SFO_API void sfo_bn_act(float *x, int n, int c, const float *gamma, const float *beta,
                        const float *mean, const float *var, float eps, const float *residual,
                        int relu) {
  for (int i = 0; i < n; ++i) {
    for (int q = 0; q < c; ++q) {
      size_t at = (size_t)i * c + q;
      float inv = 1.0f / sqrtf(var[q] + eps);
      float y = (x[at] - mean[q]) * inv * gamma[q] + beta[q];
      if (residual) y += residual[at];
      if (relu && !(y > 0.0f)) y = 0.0f;
      x[at] = y;
    }
  }
}

/* SparseConvTensor.dense() + view, structure.py:55-64 and
 * sparse_encoder.py:122-125: out (B, C*D, H, W), channel = c*D + z. */
SFO_API void sfo_dense_bev(const float *feat, const int *idx4, int n, int c, int B, int D, int H,
                           int W, float *out) {
  memset(out, 0, sizeof(float) * (size_t)B * c * D * H * W);
  for (int i = 0; i < n; ++i) {
    const int *p = idx4 + 4 * (size_t)i;
    for (int q = 0; q < c; ++q)
      out[((((size_t)p[0] * c + q) * D + p[1]) * H + p[2]) * W + p[3]] = feat[(size_t)i * c + q];
  }
}

/* dynamic_point_to_voxel_forward: mmdet3d/ops/voxel/src/scatter_points_cuda.cu:230-311 restated
 * sequentially.  Linear id per point with dims = per-dimension max + 1 (:241, coors_id_kernel
 * :80-95, -1 if any coordinate is negative); unique ids ascending = output rows (argsort :258,
 * segment heads :97-115, cumsum :271); out_coors = the point's coordinates (:129-133);
 * sum / mean / max over the member points (:135-148, mean divides by the count :303-305; the
 * count array stays zero for sum and max as in the reference).  Sums run in point order in fp32
 * (the reference's atomics have no fixed order).  reduce: 0 sum, 1 mean, 2 max.
 * Outputs sized for n rows; returns the number of voxels. */
static int cmp_i64(const void *a, const void *b) {
  long long x = *(const long long *)a, y = *(const long long *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

SFO_API int sfo_dynamic_scatter(const float *feats, const int *coors, int n, int c, int ndim,
                                int reduce, float *vfeats, int *vcoors, int *p2v, int *count) {
  if (n == 0) return 0;
  long long dims[8];
  for (int i = 0; i < ndim; ++i) {
    int m = -1;
    for (int x = 0; x < n; ++x)
      if (coors[(size_t)x * ndim + i] > m) m = coors[(size_t)x * ndim + i];
    dims[i] = (long long)m + 1;
  }
  long long *ids = (long long *)malloc(sizeof(long long) * (size_t)n);
  long long *uniq = (long long *)malloc(sizeof(long long) * (size_t)n);
  int m = 0;
  for (int x = 0; x < n; ++x) {
    long long id = 0;
    for (int i = 0; i < ndim && id != -1; ++i) {
      long long t = coors[(size_t)x * ndim + i];
      id = (t < 0) ? -1 : id * dims[i] + t;
    }
    ids[x] = id;
    if (id >= 0) uniq[m++] = id;
  }
  qsort(uniq, (size_t)m, sizeof(long long), cmp_i64);
  int nv = 0;
  for (int i = 0; i < m; ++i)
    if (i == 0 || uniq[i] != uniq[i - 1]) uniq[nv++] = uniq[i];
  for (int v = 0; v < nv; ++v) {
    count[v] = 0;
    for (int q = 0; q < c; ++q) vfeats[(size_t)v * c + q] = (reduce == 2) ? -INFINITY : 0.0f;
  }
  int *members = (int *)calloc((size_t)nv > 0 ? (size_t)nv : 1, sizeof(int));
  for (int x = 0; x < n; ++x) {
    if (ids[x] < 0) {
      p2v[x] = -1;
      continue;
    }
    int lo = 0, hi = nv - 1;
    while (lo < hi) {
      int mid = (lo + hi) / 2;
      if (uniq[mid] < ids[x]) lo = mid + 1; else hi = mid;
    }
    p2v[x] = lo;
    members[lo]++;
    for (int i = 0; i < ndim; ++i) vcoors[(size_t)lo * ndim + i] = coors[(size_t)x * ndim + i];
    for (int q = 0; q < c; ++q) {
      float f = feats[(size_t)x * c + q], *o = &vfeats[(size_t)lo * c + q];
      if (reduce == 2) *o = fmaxf(f, *o); else *o = *o + f;
    }
  }
  if (reduce == 1)
    for (int v = 0; v < nv; ++v) {
      count[v] = members[v];
      for (int q = 0; q < c; ++q) vfeats[(size_t)v * c + q] /= (float)members[v];
    }
  free(members);
  free(uniq);
  free(ids);
  return nv;
}

SFO_API void sfo_set_num_threads(int n) {
#ifdef _OPENMP
  if (n >= 1) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

SFO_API int sfo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

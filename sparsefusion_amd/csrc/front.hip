// The coordinate phase of a frame -- voxelizer, coordinate hashes, every rulebook of the encoder
// and the plans of the mask-ordered layers -- enqueued by ONE call, sf_lidar_front.
//
// Reference: what SparseFusionDetector.voxelize (mmdet3d/models/detectors/sparsefusion.py:133-161)
// and the get_indice_pairs call of every SparseConvolution.forward (mmdet3d/ops/spconv/conv.py:
// 113-204 -> spconv_ops.h:27-141) do, spread over a frame.  The stand-alone entry points
// (sf_hard_voxelize, sf_hash_build, sf_neighbor_table, sf_strided_rulebook, sf_row_order_by_mask)
// give the same buffers; this call runs the same kernels with fewer launches, which is what the
// phase costs on MI355X (a launch is 4-5 us whatever it does, and the stand-alone sequence has
// 52 of them per nuScenes frame):
//   * one fill kernel for every table of the frame (slot tables of the voxelizer, hash keys,
//     first-touch words, histogram tables, counters) instead of nine;
//   * the launch that builds the voxels' coordinate hash also opens (-1) the first SubM table, and
//     one launch per strided rulebook opens its own table and the SubM table of its output set
//     (folding these stores into the row-assignment kernels was measured: 14 -> 41 us for the
//     assignment, a thread per INPUT writes its rows' 40 entries one after the other);
//   * the table scatter of a strided rulebook and the SubM table of its output set run in one
//     launch (both wait for the row assignment only).
#include <string.h>

#include "front_internal.h"

namespace {

constexpr int kMaxFill = 40;
struct FillJob {
  uint32_t *p;
  unsigned long long words;
  uint32_t even, odd;   // value of the even / odd 32-bit words
};
struct FillJobs {
  int n;
  FillJob job[kMaxFill];
};

// every job in turn, all workgroups on each: 16-byte stores (buffers are 16-byte aligned: checked)
__global__ void __launch_bounds__(256) fill_many_kernel(FillJobs fj) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthr = (size_t)gridDim.x * 256;
  for (int j = 0; j < fj.n; ++j) {
    const FillJob &f = fj.job[j];
    uint4 *p4 = reinterpret_cast<uint4 *>(f.p);
    const uint4 v = make_uint4(f.even, f.odd, f.even, f.odd);
    const size_t n4 = f.words >> 2;
    for (size_t i = tid; i < n4; i += nthr) p4[i] = v;
    const size_t tail = f.words & 3;
    if (tid < tail) f.p[(n4 << 2) + tid] = (tid & 1) ? f.odd : f.even;
  }
}

bool is_subm27(const SfRulebook &r) {
  return r.subm && r.src == r.dst && r.ksize[0] == 3 && r.ksize[1] == 3 && r.ksize[2] == 3;
}

}  // namespace


extern "C" int sf_lidar_front(const SfFrontDesc *d, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  SF_CHECK_ARG(d != nullptr, "descriptor is NULL");
  SF_CHECK_ARG(d->batch >= 1 && d->points && d->num_points && d->num_features >= 3,
               "bad batch / points / num_points / num_features");
  SF_CHECK_ARG(d->n_sets >= 1 && d->sets && d->counts_dev, "coordinate sets / counts are NULL");
  SF_CHECK_ARG(d->n_rulebooks >= 0 && (d->n_rulebooks == 0 || d->rulebooks), "rulebooks is NULL");
  SF_CHECK_ARG(d->n_rulebooks <= 64, "more than 64 rulebooks");
  const SfCoordSet &s0 = d->sets[0];
  SF_CHECK_ARG(s0.indices && s0.keys && s0.vals && s0.capacity >= 4 &&
                   (s0.capacity & (s0.capacity - 1)) == 0 && s0.capacity >= s0.bound,
               "coordinate set 0: buffers NULL or capacity %d not a power of two >= bound %d",
               s0.capacity, s0.bound);
  int max_n = 0;
  for (int b = 0; b < d->batch; ++b) {
    SF_CHECK_ARG(d->num_points[b] >= 0 && (d->num_points[b] == 0 || d->points[b]), "sample %d: points", b);
    if (d->num_points[b] > max_n) max_n = d->num_points[b];
  }
  const size_t vws = sf_hard_voxelize_workspace(max_n, d->max_points);
  SF_CHECK_ARG(d->max_points >= 1 && vws > 0, "max_points=%d", d->max_points);
  if (!d->vox_workspace || d->vox_workspace_bytes < vws) {
    sf_set_error("voxelizer workspace: %zu bytes per sample given, %zu needed", d->vox_workspace_bytes, vws);
    return SF_ERR_WORKSPACE;
  }
  int *overflow_dev = d->counts_dev + d->n_sets;

  // ---- one fill for every table of the frame ----------------------------------------------------
  FillJobs fj;
  fj.n = 0;
  auto add = [&](void *p, size_t words, uint32_t even, uint32_t odd) -> int {
    if (words == 0) return SF_OK;
    SF_CHECK_ARG(p != nullptr && ((uintptr_t)p & 15) == 0, "a table of the frame is NULL or not 16-byte aligned");
    if (fj.n == kMaxFill) {   // flush (does not happen with the encoder's eight rulebooks)
      hipLaunchKernelGGL(fill_many_kernel, dim3(1024), dim3(256), 0, st, fj);
      SF_CHECK_LAUNCH();
      fj.n = 0;
    }
    fj.job[fj.n++] = FillJob{(uint32_t *)p, (unsigned long long)words, even, odd};
    return SF_OK;
  };
  int rc = add(d->counts_dev, (size_t)d->n_sets + 1, 0u, 0u);
  if (rc != SF_OK) return rc;
  for (int b = 0; b < d->batch; ++b) {
    if (d->num_points[b] == 0) continue;
    uint32_t *tab;
    size_t words;
    sf_int_vox_table((char *)d->vox_workspace + (size_t)b * d->vox_workspace_bytes, d->num_points[b],
                     d->max_points, &tab, &words);
    if ((rc = add(tab, words, SF_EMPTY_KEY, (uint32_t)SF_INT_MAX)) != SF_OK) return rc;
  }
  if ((rc = add(s0.keys, (size_t)s0.capacity, SF_EMPTY_KEY, SF_EMPTY_KEY)) != SF_OK) return rc;
  if ((rc = add(s0.vals, (size_t)s0.capacity, (uint32_t)SF_INT_MAX, (uint32_t)SF_INT_MAX)) != SF_OK) return rc;
  for (int i = 0; i < d->n_rulebooks; ++i) {
    const SfRulebook &r = d->rulebooks[i];
    SF_CHECK_ARG(r.src >= 0 && r.src < d->n_sets && r.dst >= 0 && r.dst < d->n_sets && r.nbr,
                 "rulebook %d: bad set index or NULL table", i);
    if (!r.subm) {
      const SfCoordSet &o = d->sets[r.dst];
      SF_CHECK_ARG(r.dst != r.src && r.first && o.keys && o.vals && o.indices && o.capacity >= 4 &&
                       (o.capacity & (o.capacity - 1)) == 0,
                   "strided rulebook %d: output set buffers / first-touch scratch", i);
      if ((rc = add(o.keys, (size_t)o.capacity, SF_EMPTY_KEY, SF_EMPTY_KEY)) != SF_OK) return rc;
      if ((rc = add(r.first, (size_t)o.capacity, (uint32_t)SF_INT_MAX, (uint32_t)SF_INT_MAX)) != SF_OK) return rc;
    }
    if (r.plan) {
      SF_CHECK_ARG(r.plan_ws != nullptr, "rulebook %d: plan without workspace", i);
      uint32_t *h;
      size_t words;
      sf_int_row_order_hist(r.plan_ws, d->sets[r.dst].bound, &h, &words);
      if ((rc = add(h, words, 0u, 0u)) != SF_OK) return rc;
    }
  }
  hipLaunchKernelGGL(fill_many_kernel, dim3(1024), dim3(256), 0, st, fj);
  SF_CHECK_LAUNCH();

  // ---- voxelizer: rows of every sample appended ----------------------------------------------------
  SfVoxFuse vf;
  vf.tables_ready = 1;
  bool done[64] = {};
  bool prefilled[64] = {};
  {
    long long vol = (long long)d->batch * s0.shape[0] * s0.shape[1] * s0.shape[2];
    SF_CHECK_ARG(vol > 0 && vol < 0xFFFFFFFFll, "batch*D*H*W of set 0 does not fit 32-bit keys");
  }
  {   // every sample through the same launches (blockIdx.y = sample), kVoxBatch at a time
    SF_CHECK_ARG(d->batch <= 64, "batch %d: at most 64 samples per call", d->batch);
    void *wss[64];
    for (int b = 0; b < d->batch; ++b) wss[b] = (char *)d->vox_workspace + (size_t)b * d->vox_workspace_bytes;
    rc = sf_int_hard_voxelize(d->batch, d->points, d->num_points, d->num_features, d->voxel_size,
                              d->coors_range, d->max_points, d->max_voxels, d->voxels, s0.indices, 4, 0,
                              d->num_points_per_voxel, d->mean, d->mean_features, d->mean_stride,
                              d->counts_dev, s0.bound, overflow_dev, wss, d->vox_workspace_bytes, st, &vf);
    if (rc != SF_OK) return rc;
  }
  // ---- coordinate hash of the voxels; the same launch opens the first SubM table ---------------------
  {
    SfNbrPrefill pf = {nullptr, 0, 0, 0};
    for (int i = 0; i < d->n_rulebooks; ++i) {
      const SfRulebook &r = d->rulebooks[i];
      if (is_subm27(r) && r.src == 0 && r.nbr_stride >= s0.bound) {
        pf = SfNbrPrefill{r.nbr, r.nbr_stride, 14, 27};
        prefilled[i] = true;
        break;
      }
    }
    // (vals need the INT_MAX fill: atomicMin keeps the first of duplicate coordinates)
    rc = sf_int_hash_build(s0.indices, d->counts_dev, s0.bound, s0.shape, d->batch, s0.keys, s0.vals,
                           s0.capacity, overflow_dev, 1, pf, st);
    if (rc != SF_OK) return rc;
  }

  // ---- rulebooks in order ------------------------------------------------------------------------
  auto plan_of = [&](int i) -> int {
    const SfRulebook &r = d->rulebooks[i];
    if (!r.plan) return SF_OK;
    int K = r.ksize[0] * r.ksize[1] * r.ksize[2];
    return sf_int_row_order(r.nbr, r.nbr_stride, K, r.subm ? (K - 1) / 2 : -1, d->counts_dev + r.dst,
                            d->sets[r.dst].bound, r.plan,
                            r.plan_ws, r.plan_ws_bytes, 1, st);
  };
  for (int i = 0; i < d->n_rulebooks; ++i) {
    if (done[i]) continue;
    const SfRulebook &r = d->rulebooks[i];
    const SfCoordSet &src = d->sets[r.src], &dst = d->sets[r.dst];
    if (r.subm) {
      if (is_subm27(r)) {
        rc = sf_int_subm27(dst.indices, d->counts_dev + r.dst, dst.bound, d->batch, src.shape, src.keys,
                           src.vals, src.capacity, r.nbr, r.nbr_stride, prefilled[i] ? 1 : 0, nullptr, st);
      } else {
        int pad[3] = {r.ksize[0] / 2, r.ksize[1] / 2, r.ksize[2] / 2};
        int one[3] = {1, 1, 1};
        int K = r.ksize[0] * r.ksize[1] * r.ksize[2];
        rc = sf_neighbor_table(dst.indices, d->counts_dev + r.dst, dst.bound, d->batch, src.shape,
                               r.ksize, one, pad, src.keys, src.vals, src.capacity, K % 2, r.nbr,
                               r.nbr_stride, stream);
      }
      if (rc != SF_OK) return rc;
      done[i] = true;
      if ((rc = plan_of(i)) != SF_OK) return rc;
      continue;
    }
    // strided: look ahead for the SubM 3x3x3 table of the output set
    int nx = -1;
    for (int j = i + 1; j < d->n_rulebooks; ++j)
      if (!done[j] && is_subm27(d->rulebooks[j]) && d->rulebooks[j].src == r.dst &&
          d->rulebooks[j].nbr_stride >= dst.bound) {
        nx = j;
        break;
      }
    SfStridedFuse sfz;
    memset(&sfz, 0, sizeof(sfz));
    sfz.tables_ready = 1;
    sfz.first = r.first;
    SfScatterJob job;
    if (nx >= 0) {
      sfz.pf_next = SfNbrPrefill{d->rulebooks[nx].nbr, d->rulebooks[nx].nbr_stride, 14, 27};
      sfz.defer_scatter = 1;
    }
    rc = sf_int_strided_rulebook(src.indices, d->counts_dev + r.src, src.bound, d->batch, dst.shape,
                                 r.ksize, r.stride, r.padding, dst.keys, dst.vals, dst.capacity,
                                 dst.indices, dst.bound, d->counts_dev + r.dst, overflow_dev, r.nbr,
                                 r.nbr_stride, d->rb_workspace, d->rb_workspace_bytes, st, &sfz, &job);
    if (rc != SF_OK) return rc;
    done[i] = true;
    if (nx >= 0) {
      const SfRulebook &q = d->rulebooks[nx];
      rc = sf_int_subm27(dst.indices, d->counts_dev + q.dst, dst.bound, d->batch, dst.shape, dst.keys,
                         dst.vals, dst.capacity, q.nbr, q.nbr_stride, 1, &job, st);
      if (rc != SF_OK) return rc;
      done[nx] = true;
    }
    if ((rc = plan_of(i)) != SF_OK) return rc;
    if (nx >= 0 && (rc = plan_of(nx)) != SF_OK) return rc;
  }
  return SF_OK;
}

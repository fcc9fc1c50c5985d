// Hard / dynamic voxelization for gfx950.
//
// Reference semantics (mmdet3d/ops/voxel/src/voxelization_cpu.cpp:43-99, the CUDA path
// voxelization_cuda.cu:184-326 computes the same result with an O(N^2) search and a serial
// <<<1,1>>> kernel): voxels are numbered in first-point order, the scan stops at the first point
// that would open voxel number max_voxels, and each voxel keeps its first max_points points.
//
// MI355X formulation (all integer / byte work, HBM- and atomic-bound, no serial step):
//   fill    : hash slots (key, first point) <- (EMPTY, INT_MAX)              [8 bytes per slot]
//   insert  : per point: coordinate -> hash slot (atomicCAS on the key), atomicMin of the point
//             index into the slot's `first` word, which sits next to the key (one 64-byte line
//             serves both); the point remembers its slot.  Consecutive points of one voxel (a
//             LiDAR ring sweeping over a near surface) form a RUN inside a wave: only the head of
//             a run offers its index, and the max_points-th point of a run posts its index as the
//             voxel's BOUND (at least max_points points of the voxel lie at or below it)
//   count   : flag "point i is the first point of its voxel" (first[slot] == i), block sums
//   assign  : exclusive scan of the flags = voxel id in first-point order; the point whose
//             voxel id equals max_voxels is the cut (reference `break`); writes coors, slot -> voxel
//             id, and opens the voxel's point list (INT_MAX) in a table indexed by VOXEL ID
//   collect : per point below the cut whose voxel survived (and, where the voxel has a bound, at
//             or below it: a voxel that swallows 30 000 points of a frame -- a return right at the
//             sensor -- otherwise has them all cascade through one 40-byte list at once, measured
//             4.2 ms for one launch): bounded sorted insert of the point
//             index into the voxel's list of the max_points SMALLEST indices (a cascade of
//             atomicMin: an entry keeps the min of everything offered to it and passes the rest on;
//             the multiset is conserved, so the result is order independent)
//   write   : voxels[v][s][:] = points[list[v][s]], zeros elsewhere
//   reduce  : num_points_per_voxel and the fused HardSimpleVFE mean
// The lists are dense by voxel id (max_voxels x max_points ints, L2 resident) and built only for the
// voxels that survive the cut; round 1 kept a list per HASH SLOT (21 MB filled and probed at random
// per frame: 146 MB at the fabric for a 6.4 MB cloud).
#include <mutex>

#include "front_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kItems = 8;                    // points per thread in the scan kernels
constexpr int kChunk = kThreads * kItems;    // points per block in the scan kernels

struct VoxParams {
  float vs[3];
  float lo[3];
  int grid[3];  // x, y, z
};

// voxelization_cpu.cpp:20-37 / voxelization_cuda.cu:33-55.  IEEE subtract, IEEE divide, floor.
// The comparison is done on the float so NaN / Inf / huge values are rejected explicitly (the
// reference reaches the same outcome through an out-of-range float->int cast).
__device__ __forceinline__ bool point_coor(const float *__restrict__ p, const VoxParams &vp,
                                           int c_zyx[3]) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float v = floorf((p[j] - vp.lo[j]) / vp.vs[j]);
    if (!(v >= 0.0f && v < (float)vp.grid[j])) return false;
    c_zyx[2 - j] = (int)v;
  }
  return true;
}

__global__ void dynamic_voxelize_kernel(const float *__restrict__ points, int n, int f,
                                        VoxParams vp, int *__restrict__ coors) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int c[3];
    bool ok = point_coor(points + (size_t)i * f, vp, c);
    coors[3 * (size_t)i + 0] = ok ? c[0] : -1;
    coors[3 * (size_t)i + 1] = ok ? c[1] : -1;
    coors[3 * (size_t)i + 2] = ok ? c[2] : -1;
  }
}

// One sample of a launch: its points and its slice of the workspace.  Every kernel of the hard path
// takes up to kVoxBatch samples (blockIdx.y picks one): the samples of a batch are independent but
// for the base row of their voxels, and each of these kernels is latency bound at one frame's size,
// so a batch costs little more than one sample (8 samples launched one after the other: 8 x).
struct VoxSample {
  const float *points;
  uint32_t *kf, *sbound;
  int *lists, *pslot, *bsum, *misc, *slot_vid;
  int n, batch_idx, nblk;
  uint32_t mask;
};
constexpr int kVoxBatch = 8;
struct VoxBatch {
  VoxSample s[kVoxBatch];
};

// misc words: [0] cut point index, [1] voxel_num of this sample, [2] overflow, [3] base row of the
// sample's voxels, [4] row count in front of the launch's first sample
// slot table: kf[2*slot] = key, kf[2*slot + 1] = smallest point index of the voxel; the bound words
// (one per slot, unsigned: both fill values read as "no bound") follow the table and share its fill
__global__ void vox_fill_kernel(uint2 *__restrict__ kf, long long n_slots, int *__restrict__ misc,
                                const int *__restrict__ row_count_dev) {
  long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  const uint4 init = make_uint4(SF_EMPTY_KEY, (uint32_t)SF_INT_MAX, SF_EMPTY_KEY, (uint32_t)SF_INT_MAX);
  uint4 *kf4 = reinterpret_cast<uint4 *>(kf);
  for (long long i = tid; i < n_slots / 2; i += stride) kf4[i] = init;   // n_slots is a power of two
  if (tid == 0) {
    misc[0] = SF_INT_MAX;
    misc[1] = 0;
    misc[2] = 0;
    misc[3] = row_count_dev ? *row_count_dev : 0;
  }
}

__global__ void vox_insert_kernel(VoxBatch vb, int f, int max_points, VoxParams vp,
                                  int *__restrict__ overflow_dev) {
  const VoxSample &S = vb.s[blockIdx.y];
  const float *__restrict__ points = S.points;
  uint32_t *__restrict__ kf = S.kf, *__restrict__ sbound = S.sbound;
  int *__restrict__ pslot = S.pslot;
  const int n = S.n;
  const uint32_t mask = S.mask;
  const int lane = threadIdx.x & 63;
  // (the loop bound is the same for the lanes of a wave: the run detection below shuffles)
  for (int base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
    const int i = base + threadIdx.x;
    int c[3];
    uint32_t key = SF_EMPTY_KEY;   // never a voxel's key (the grid volume is below 2^32 - 1)
    if (i < n && point_coor(points + (size_t)i * f, vp, c))
      key = ((uint32_t)c[0] * (uint32_t)vp.grid[1] + (uint32_t)c[1]) * (uint32_t)vp.grid[0] + (uint32_t)c[2];
    const bool valid = key != SF_EMPTY_KEY;
    // runs of consecutive lanes in one voxel: the head probes the table for the whole run and
    // holds the run's smallest index
    const uint32_t prev = __shfl_up(key, 1, 64);
    const bool head = lane == 0 || key != prev || !valid;
    const unsigned long long heads = __ballot(head);
    int leader = 63 - __builtin_clzll(heads & ((2ull << lane) - 1ull));
    bool post = lane - leader == max_points - 1;
    // up to two voxels that continue a run in this wave are matched over the WHOLE wave (a ring that
    // alternates between two voxels, runs broken by a stray point): one probe, one offer per voxel
    // and wave, and the bound from all of the wave's lanes in the voxel
    unsigned long long multi = __ballot(valid && !head);
    for (int it = 0; it < 2 && multi; ++it) {
      const uint32_t k_hot = __shfl(key, __builtin_ctzll(multi), 64);
      const unsigned long long m = __ballot(key == k_hot);
      if (key == k_hot) {
        leader = __builtin_ctzll(m);
        post = __popcll(m & ((1ull << lane) - 1ull)) == max_points - 1;
      }
      multi &= ~m;
    }
    int found = -1;
    if (valid && leader == lane) {
      uint32_t slot = sf_scatter_slot(key, mask);
      for (uint32_t probe = 0; probe <= mask; ++probe) {
        uint32_t cur = kf[2 * (size_t)slot];
        if (cur == SF_EMPTY_KEY) {
          cur = atomicCAS(&kf[2 * (size_t)slot], SF_EMPTY_KEY, key);
          if (cur == SF_EMPTY_KEY) cur = key;
        }
        if (cur == key) {
          found = (int)slot;
          break;
        }
        slot = (slot + 1) & mask;
      }
      if (found < 0 && overflow_dev) *overflow_dev = 1;  // cannot happen with capacity >= n; a guard
    }
    found = __shfl(found, leader, 64);
    if (!valid) found = -1;
    if (i < n) pslot[i] = found;
    if (found >= 0) {
      if (leader == lane) atomicMin(reinterpret_cast<int *>(kf) + 2 * (size_t)found + 1, i);
      if (post) atomicMin(&sbound[found], (uint32_t)i);
    }
  }
}

__device__ __forceinline__ bool is_first_point(int i, int n, const int *__restrict__ pslot,
                                               const uint32_t *__restrict__ kf, int *slot_out) {
  if (i >= n) return false;
  int slot = pslot[i];
  *slot_out = slot;
  if (slot < 0) return false;
  return (int)kf[2 * (size_t)slot + 1] == i;
}

constexpr int kBoundedBit = 1 << 30;

__global__ void __launch_bounds__(kThreads)
vox_count_kernel(VoxBatch vb, const int *__restrict__ row_count_dev) {
  __shared__ int smem[17];
  const VoxSample &S = vb.s[blockIdx.y];
  if ((int)blockIdx.x >= S.nblk) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) {   // read by vox_assign, the next launch
    S.misc[0] = SF_INT_MAX;
    S.misc[1] = 0;
    S.misc[4] = row_count_dev ? *row_count_dev : 0;
  }
  int base = blockIdx.x * kChunk + threadIdx.x * kItems;
  int c = 0;
#pragma unroll
  for (int e = 0; e < kItems; ++e) {
    int slot;
    c += is_first_point(base + e, S.n, S.pslot, S.kf, &slot) ? 1 : 0;
  }
  int total = sf_block_sum(c, smem);
  if (threadIdx.x == 0) S.bsum[blockIdx.x] = total;
}

// voxels a sample contributes: its first points, cut at max_voxels, clipped to the rows left
__device__ __forceinline__ int vox_rows_of(int all, int max_voxels, int base_row, int rows_capacity,
                                           bool *clipped) {
  int vnum = (max_voxels >= 0 && all > max_voxels) ? max_voxels : all;
  *clipped = base_row + vnum > rows_capacity;
  if (*clipped) vnum = rows_capacity > base_row ? rows_capacity - base_row : 0;
  return vnum;
}

__global__ void __launch_bounds__(kThreads)
vox_assign_kernel(VoxBatch vb, int count, int max_points, int max_voxels, VoxParams vp,
                  int *__restrict__ coors, int coors_stride, int rows_capacity,
                  int *__restrict__ row_count_dev, int *__restrict__ overflow_dev) {
  __shared__ int smem[17];
  const int sample = blockIdx.y;
  const VoxSample &S = vb.s[sample];
  if ((int)blockIdx.x >= S.nblk) return;
  const int n = S.n;
  const int *__restrict__ pslot = S.pslot;
  const uint32_t *__restrict__ kf = S.kf;
  int *__restrict__ misc = S.misc;
  // base row: the samples in front of this one in the launch append first (same rule as a
  // sequence of single-sample calls: cut at max_voxels, clipped to the rows that are left)
  int base_row = misc[4];
  for (int q = 0; q < sample; ++q) {
    int pa = 0;
    for (int i = threadIdx.x; i < vb.s[q].nblk; i += blockDim.x) pa += vb.s[q].bsum[i];
    const int all_q = sf_block_sum(pa, smem);
    bool clipped;
    base_row += vox_rows_of(all_q, max_voxels, base_row, rows_capacity, &clipped);
  }
  int before, all;
  sf_block_prefix_of(S.bsum, S.nblk, blockIdx.x, smem, &before, &all);
  int base = blockIdx.x * kChunk + threadIdx.x * kItems;
  int slots[kItems];
  unsigned flags = 0;
  int c = 0;
#pragma unroll
  for (int e = 0; e < kItems; ++e) {
    int slot = -1;
    bool f = is_first_point(base + e, n, pslot, kf, &slot);
    slots[e] = slot;
    flags |= (f ? 1u : 0u) << e;
    c += f ? 1 : 0;
  }
  int total;
  int vid = before + sf_block_excl_scan(c, smem, &total);
#pragma unroll
  for (int e = 0; e < kItems; ++e) {
    if (!((flags >> e) & 1u)) continue;
    if (max_voxels >= 0 && vid == max_voxels) misc[0] = base + e;  // the reference's `break`
    const bool keep = max_voxels < 0 || vid < max_voxels;
    // bit 30: the voxel has a bound (vox_collect then reads it; the fill values have bits 0-30 set)
    const bool bounded = (S.sbound[slots[e]] & 0x7FFFFFFFu) != 0x7FFFFFFFu;
    S.slot_vid[slots[e]] = keep ? (vid | (bounded ? kBoundedBit : 0)) : -1;
    if (keep) {
      int *list = S.lists + (size_t)vid * max_points;
      list[0] = base + e;   // the first point is the smallest index: it opens the list
      for (int s = 1; s < max_points; ++s) list[s] = SF_INT_MAX;
      int row = base_row + vid;
      if (row < rows_capacity) {
        uint32_t key = kf[2 * (size_t)slots[e]];
        uint32_t gx = (uint32_t)vp.grid[0], gy = (uint32_t)vp.grid[1];
        int x = (int)(key % gx);
        uint32_t zy = key / gx;
        int y = (int)(zy % gy), z = (int)(zy / gy);
        int *o = coors + (size_t)row * coors_stride;
        if (coors_stride == 4) *o++ = S.batch_idx;
        o[0] = z;
        o[1] = y;
        o[2] = x;
      } else if (overflow_dev) {
        *overflow_dev = 1;
      }
    }
    ++vid;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    bool clipped;
    const int vnum = vox_rows_of(all, max_voxels, base_row, rows_capacity, &clipped);
    if (clipped && overflow_dev) *overflow_dev = 1;
    misc[1] = vnum;
    misc[3] = base_row;
    if (row_count_dev && sample == count - 1) *row_count_dev = base_row + vnum;
  }
}

// second pass over the points: the lists of the surviving voxels.
// MP > 0: the list length is a compile-time constant and the WHOLE list is read at once (one round
// trip instead of one per entry: the kernel's time is set by the crowded voxels, whose points walked
// up to max_points dependent loads before their first atomic); the insert position is the number of
// entries smaller than the point (entries only ever decrease, so a stale read can only start the
// cascade too early, never too late).  MP == 0: any max_points, entry by entry.
template <int MP>
__global__ void vox_collect_kernel(VoxBatch vb, int max_points) {
  const VoxSample &S = vb.s[blockIdx.y];
  const int n = S.n;
  const int *__restrict__ pslot = S.pslot, *__restrict__ slot_vid = S.slot_vid;
  const uint32_t *__restrict__ sbound = S.sbound;
  int *__restrict__ lists = S.lists;
  const int cut = S.misc[0];
  const int stop = cut < n ? cut : n;   // points from the cut on belong to no voxel
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < stop; i += gridDim.x * blockDim.x) {
    const int slot = pslot[i];
    if (slot < 0) continue;
    const int sv = slot_vid[slot];
    if (sv < 0) continue;
    // max_points points of the voxel lie at or below its bound: this one cannot be among them
    if ((sv & kBoundedBit) && (uint32_t)i > sbound[slot]) continue;
    const int vid = sv & (kBoundedBit - 1);
    int *list = lists + (size_t)vid * max_points;
    int v = i;
    int s0 = 0;
    if constexpr (MP > 0) {
      int e[MP];
#pragma unroll
      for (int s = 0; s < MP; ++s) e[s] = __builtin_nontemporal_load(list + s);
      if (e[0] == v) continue;          // the voxel's first point: in place since vox_assign
      int p = 0;
#pragma unroll
      for (int s = 0; s < MP; ++s) p += e[s] < v ? 1 : 0;
      if (p >= MP) continue;            // the list already holds max_points smaller indices
      s0 = p;
    } else {
      // crowded voxels: once the list holds max_points smaller indices this point has no place in it
      if (__builtin_nontemporal_load(list + max_points - 1) < v) continue;
    }
    for (int s = s0; s < max_points; ++s) {
      if constexpr (MP == 0) {
        // a (possibly stale) smaller value means "not ours"
        int cur = __builtin_nontemporal_load(list + s);
        if (cur < v) continue;
        if (cur == v) break;   // the voxel's first point: already in place (vox_assign)
      }
      int old = atomicMin(list + s, v);
      if (old > v) v = old;  // we took the entry, carry the displaced value on
      if (v == SF_INT_MAX) break;
    }
  }
}

__global__ void vox_write_kernel(VoxBatch vb, int f, int max_points, float *__restrict__ voxels) {
  const VoxSample &S = vb.s[blockIdx.y];
  const float *__restrict__ points = S.points;
  const int *__restrict__ lists = S.lists;
  const int vnum = S.misc[1], base_row = S.misc[3];
  const long long total = (long long)vnum * max_points * f;
  const int per_voxel = max_points * f;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    int vid = (int)(e / per_voxel);
    int r = (int)(e - (long long)vid * per_voxel);
    int s = r / f, k = r - s * f;
    int j = lists[(size_t)vid * max_points + s];
    float v = (j != SF_INT_MAX) ? points[(size_t)j * f + k] : 0.0f;
    voxels[(size_t)(base_row + vid) * per_voxel + r] = v;
  }
}

// num_points_per_voxel (voxelization_cpu.cpp:89-95) + HardSimpleVFE mean (voxel_encoder.py:42-44:
// sum over the slots in order, zeros included, divided by the count).
__global__ void vox_reduce_kernel(VoxBatch vb, int f, int max_points, int *__restrict__ npv,
                                  float *__restrict__ mean, int mean_features, int mean_stride) {
  const VoxSample &S = vb.s[blockIdx.y];
  const float *__restrict__ points = S.points;
  const int *__restrict__ lists = S.lists;
  const int vnum = S.misc[1], base_row = S.misc[3];
  const int cols = mean ? mean_stride : 1;
  const long long total = (long long)vnum * cols;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    int vid = (int)(e / cols), k = (int)(e - (long long)vid * cols);
    const int *list = lists + (size_t)vid * max_points;
    int cnt = 0;
    float sum = 0.0f;
    const bool want = mean && k < mean_features;
    for (int s = 0; s < max_points; ++s) {
      int j = list[s];
      if (j == SF_INT_MAX) break;  // lists ascend; INT_MAX marks empty
      ++cnt;
      if (want) sum += points[(size_t)j * f + k];
    }
    if (k == 0 && npv) npv[base_row + vid] = cnt;
    if (mean) mean[(size_t)(base_row + vid) * mean_stride + k] = want ? sum / (float)cnt : 0.0f;
  }
}

// vox_write + vox_reduce in one launch: a thread owns column k of one voxel (k < max(f, mean
// stride)), reads the voxel's list once, copies its column of every listed point and sums it in
// slot order as it goes (the mean adds the same values in the same order as vox_reduce_kernel).
template <int MP>
__global__ void vox_write_reduce_kernel(VoxBatch vb, int f, float *__restrict__ voxels,
                                        int *__restrict__ npv, float *__restrict__ mean,
                                        int mean_features, int mean_stride) {
  const VoxSample &S = vb.s[blockIdx.y];
  const float *__restrict__ points = S.points;
  const int *__restrict__ lists = S.lists;
  const int vnum = S.misc[1], base_row = S.misc[3];
  const int cols = (mean && mean_stride > f) ? mean_stride : f;
  const long long total = (long long)vnum * cols;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int vid = (int)(e / cols), k = (int)(e - (long long)vid * cols);
    const int *list = lists + (size_t)vid * MP;
    int j[MP];
#pragma unroll
    for (int s = 0; s < MP; ++s) j[s] = list[s];
    float v[MP];
#pragma unroll
    for (int s = 0; s < MP; ++s) v[s] = (k < f && j[s] != SF_INT_MAX) ? points[(size_t)j[s] * f + k] : 0.0f;
    int cnt = 0;
    float sum = 0.0f;
#pragma unroll
    for (int s = 0; s < MP; ++s) {
      if (j[s] != SF_INT_MAX) {   // lists ascend; INT_MAX marks empty
        ++cnt;
        sum += v[s];
      }
    }
    const size_t row = (size_t)(base_row + vid);
    if (k < f) {
      float *dst = voxels + row * MP * f + k;
#pragma unroll
      for (int s = 0; s < MP; ++s) dst[(size_t)s * f] = v[s];
    }
    if (k == 0 && npv) npv[row] = cnt;
    if (mean && k < mean_stride) mean[row * mean_stride + k] = k < mean_features ? sum / (float)cnt : 0.0f;
  }
}

// Standalone HardSimpleVFE (voxel_encoder.py:42-44) on an already materialised voxel tensor.
__global__ void voxel_mean_kernel(const float *__restrict__ voxels, const int *__restrict__ npv,
                                  const int *m_dev, int m_bound, int max_points, int f, int nf,
                                  float *__restrict__ out, int out_stride) {
  int m = m_dev ? *m_dev : m_bound;
  if (m > m_bound) m = m_bound;
  const long long total = (long long)m * out_stride;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    int v = (int)(e / out_stride), k = (int)(e - (long long)v * out_stride);
    float r = 0.0f;
    if (k < nf) {
      float sum = 0.0f;
      const float *base = voxels + (size_t)v * max_points * f + k;
      for (int s = 0; s < max_points; ++s) sum += base[(size_t)s * f];
      r = sum / (float)npv[v];
    }
    out[e] = r;
  }
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct VoxWorkspace {
  uint32_t *kf;        // (key, first point) per hash slot
  uint32_t *sbound;    // bound per hash slot, directly behind kf (one fill covers both)
  int *lists, *pslot, *bsum, *misc, *slot_vid;
  long long cap;
  int nblk;
  size_t bytes;
};

VoxWorkspace carve(void *ws, int n, int max_points) {
  VoxWorkspace w;
  long long want = (long long)n + n / 2;
  if (want < 1024) want = 1024;
  long long cap = 1024;
  while (cap < want) cap <<= 1;
  w.cap = cap;
  w.nblk = sf_div_up(n > 0 ? n : 1, kChunk);
  const size_t nn = (size_t)(n > 0 ? n : 1);
  size_t off = 0;
  char *b = (char *)ws;
  w.kf = (uint32_t *)(b + off);
  off += (size_t)cap * 8;              // (cap is a power of two >= 1024: no padding in between)
  w.sbound = (uint32_t *)(b + off);
  off += align_up((size_t)cap * 4);
  w.slot_vid = (int *)(b + off);
  off += align_up((size_t)cap * 4);
  w.lists = (int *)(b + off);          // by voxel id: at most one voxel per point
  off += align_up(nn * (size_t)max_points * 4);
  w.pslot = (int *)(b + off);
  off += align_up(nn * 4);
  w.bsum = (int *)(b + off);
  off += align_up((size_t)w.nblk * 4);
  w.misc = (int *)(b + off);
  off += align_up(64);
  w.bytes = off;
  return w;
}

int make_params(const float vs[3], const float range[6], VoxParams *vp) {
  SF_CHECK_ARG(vs && range, "voxel_size / coors_range is NULL");
  long long vol = 1;
  for (int i = 0; i < 3; ++i) {
    SF_CHECK_ARG(vs[i] > 0.0f, "voxel_size[%d] must be positive", i);
    vp->vs[i] = vs[i];
    vp->lo[i] = range[i];
    float q = (range[3 + i] - range[i]) / vs[i];  // fp32, voxelization_cpu.cpp:119-122
    vp->grid[i] = (int)roundf(q);
    SF_CHECK_ARG(vp->grid[i] > 0, "empty grid along axis %d", i);
    vol *= vp->grid[i];
  }
  if (vol >= 0xFFFFFFFFll) {
    sf_set_error("voxel grid volume %lld does not fit 32-bit keys", vol);
    return SF_ERR_CAPACITY;
  }
  return SF_OK;
}

}  // namespace

extern "C" int sf_dynamic_voxelize(const float *points, int n, int f, const float voxel_size[3],
                                   const float coors_range[6], int *coors, void *stream) {
  SF_CHECK_ARG(n >= 0 && f >= 3, "bad shape n=%d f=%d (need f >= 3)", n, f);
  VoxParams vp;
  int rc = make_params(voxel_size, coors_range, &vp);
  if (rc != SF_OK) return rc;
  if (n == 0) return SF_OK;
  SF_CHECK_ARG(points && coors, "points / coors is NULL");
  hipLaunchKernelGGL(dynamic_voxelize_kernel, dim3(sf_grid(n, kThreads)), dim3(kThreads), 0,
                     (hipStream_t)stream, points, n, f, vp, coors);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" size_t sf_hard_voxelize_workspace(int n, int max_points) {
  if (n < 0 || max_points < 1) return 0;
  return carve(nullptr, n, max_points).bytes;
}

namespace {

// Body of every hard-voxelize entry point: `count` samples whose voxels are appended in order
// (sample q of the call gets batch index batch_idx0 + q); workspaces[q] is device memory of at least
// carve(nullptr, n[q], max_points).bytes bytes, whoever owns it.  The samples go through the
// kernels kVoxBatch at a time.
int hard_voxelize_impl(int count, const float *const *points, const int *n, int f,
                       const float voxel_size[3], const float coors_range[6], int max_points,
                       int max_voxels, float *voxels, int *coors, int coors_stride, int batch_idx0,
                       int *num_points_per_voxel, float *mean, int mean_features, int mean_stride,
                       int *row_count_dev, int rows_capacity, int *overflow_dev, int *voxel_num_host,
                       void *const *workspaces, size_t workspace_bytes, hipStream_t st,
                       const SfVoxFuse *fuse = nullptr) {
  SF_CHECK_ARG(count >= 1 && points && n && workspaces, "bad sample list");
  SF_CHECK_ARG(f >= 3, "bad point width f=%d (need f >= 3)", f);
  SF_CHECK_ARG(max_points >= 1, "max_points=%d: the hard path needs max_points >= 1 "
               "(use sf_dynamic_voxelize for -1)", max_points);
  SF_CHECK_ARG(max_voxels >= -1, "max_voxels=%d", max_voxels);
  SF_CHECK_ARG(rows_capacity >= 0, "rows_capacity=%d", rows_capacity);
  SF_CHECK_ARG((coors || rows_capacity == 0) && (coors_stride == 3 || coors_stride == 4),
               "coors NULL or stride %d", coors_stride);
  SF_CHECK_ARG(count == 1 || row_count_dev, "several samples append through row_count_dev");
  if (mean)
    SF_CHECK_ARG(mean_features >= 1 && mean_features <= f && mean_stride >= mean_features,
                 "mean_features=%d mean_stride=%d f=%d", mean_features, mean_stride, f);
  VoxParams vp;
  int rc = make_params(voxel_size, coors_range, &vp);
  if (rc != SF_OK) return rc;
  for (int q = 0; q < count; ++q) {
    SF_CHECK_ARG(n[q] >= 0 && n[q] < (1 << 30), "sample %d: n=%d (need 0 <= n < 2^30)", q, n[q]);
    const size_t need = carve(nullptr, n[q], max_points).bytes;
    if (!workspaces[q] || workspace_bytes < need) {
      sf_set_error("workspace too small: %zu < %zu bytes", workspace_bytes, need);
      return SF_ERR_WORKSPACE;
    }
    SF_CHECK_ARG(n[q] == 0 || points[q], "points is NULL");
  }
  const bool fused_wr = voxels && (num_points_per_voxel || mean) && (max_points == 10 || max_points == 5);
  VoxWorkspace w_last = carve(workspaces[count - 1], n[count - 1], max_points);
  for (int q0 = 0; q0 < count; q0 += kVoxBatch) {
    VoxBatch vb = {};
    int nb = 0, n_max = 0, nblk_max = 0;
    for (int q = q0; q < count && q < q0 + kVoxBatch; ++q) {
      VoxWorkspace w = carve(workspaces[q], n[q], max_points);
      if (!(fuse && fuse->tables_ready && n[q] > 0)) {
        // table + bound words: 3 * cap words = (cap + cap / 2) slots' worth of fill
        hipLaunchKernelGGL(vox_fill_kernel, dim3(sf_grid(3 * w.cap / 4, kThreads)), dim3(kThreads), 0, st,
                           reinterpret_cast<uint2 *>(w.kf), w.cap + w.cap / 2, w.misc,
                           (const int *)row_count_dev);
        SF_CHECK_LAUNCH();
      }
      if (n[q] == 0) continue;   // an empty sample appends nothing
      VoxSample &S = vb.s[nb++];
      S.points = points[q];
      S.kf = w.kf;
      S.sbound = w.sbound;
      S.lists = w.lists;
      S.pslot = w.pslot;
      S.bsum = w.bsum;
      S.misc = w.misc;
      S.slot_vid = w.slot_vid;
      S.n = n[q];
      S.batch_idx = batch_idx0 + q;
      S.nblk = w.nblk;
      S.mask = (uint32_t)(w.cap - 1);
      if (n[q] > n_max) n_max = n[q];
      if (w.nblk > nblk_max) nblk_max = w.nblk;
    }
    if (nb == 0) continue;
    const dim3 blk(kThreads);
    hipLaunchKernelGGL(vox_insert_kernel, dim3(sf_grid(n_max, kThreads), nb), blk, 0, st, vb, f, max_points,
                       vp, overflow_dev);
    SF_CHECK_LAUNCH();
    hipLaunchKernelGGL(vox_count_kernel, dim3(nblk_max, nb), blk, 0, st, vb, (const int *)row_count_dev);
    SF_CHECK_LAUNCH();
    hipLaunchKernelGGL(vox_assign_kernel, dim3(nblk_max, nb), blk, 0, st, vb, nb, max_points, max_voxels, vp,
                       coors, coors_stride, rows_capacity, row_count_dev, overflow_dev);
    SF_CHECK_LAUNCH();
    auto collect = max_points == 10 ? vox_collect_kernel<10>
                   : max_points == 5 ? vox_collect_kernel<5>
                   : max_points == 20 ? vox_collect_kernel<20>
                   : max_points == 32 ? vox_collect_kernel<32> : vox_collect_kernel<0>;
    hipLaunchKernelGGL(collect, dim3(sf_grid(n_max, kThreads), nb), blk, 0, st, vb, max_points);
    SF_CHECK_LAUNCH();
    const long long bound_v = (max_voxels >= 0 && max_voxels < n_max) ? max_voxels : n_max;
    if (fused_wr) {
      const long long cols = (mean && mean_stride > f) ? mean_stride : f;
      auto kern = max_points == 10 ? vox_write_reduce_kernel<10> : vox_write_reduce_kernel<5>;
      hipLaunchKernelGGL(kern, dim3(sf_grid(bound_v * cols, kThreads), nb), blk, 0, st, vb, f, voxels,
                         num_points_per_voxel, mean, mean_features, mean_stride);
      SF_CHECK_LAUNCH();
    }
    if (voxels && !fused_wr) {
      hipLaunchKernelGGL(vox_write_kernel, dim3(sf_grid(bound_v * max_points * f, kThreads), nb), blk, 0, st,
                         vb, f, max_points, voxels);
      SF_CHECK_LAUNCH();
    }
    if ((num_points_per_voxel || mean) && !fused_wr) {
      const long long cols = mean ? mean_stride : 1;
      hipLaunchKernelGGL(vox_reduce_kernel, dim3(sf_grid(bound_v * cols, kThreads), nb), blk, 0, st, vb, f,
                         max_points, num_points_per_voxel, mean, mean_features, mean_stride);
      SF_CHECK_LAUNCH();
    }
  }
  if (voxel_num_host) {   // (single-sample entry points only)
    SF_CHECK_HIP(hipMemcpyAsync(voxel_num_host, w_last.misc + 1, sizeof(int), hipMemcpyDeviceToHost, st));
    SF_CHECK_HIP(hipStreamSynchronize(st));
  }
  return SF_OK;
}

// ---- scratch of the allocation-free entry point (sf_hard_voxelize_static) ---------------------
// One array per GPU the library is loaded on, held until the process ends.  In HOST code the
// name `scratch` is only a handle for hipGetSymbolAddress: it is not a device address and must
// never be passed to a kernel or a HIP memory call.  Host code uses the per-device pointer that
// static_scratch() returns.
constexpr size_t kStaticScratchBytes = (size_t)SF_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void *g_scratch_ptr[kMaxDevices] = {};   // device address of `scratch`, per device ordinal

// Device address of the CURRENT device's copy of `scratch`; resolved once per device (a symbol
// lookup: no allocation), then served from the table.
int static_scratch(void **ptr) {
  int dev = -1;
  SF_CHECK_HIP(hipGetDevice(&dev));
  SF_CHECK_ARG(dev >= 0 && dev < kMaxDevices, "device ordinal %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void *p = nullptr;
    SF_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    SF_CHECK_ARG(p != nullptr, "hipGetSymbolAddress returned NULL for the static scratch");
    g_scratch_ptr[dev] = p;
  }
  *ptr = g_scratch_ptr[dev];
  return SF_OK;
}

}  // namespace

int sf_int_hard_voxelize(int count, const float *const *points, const int *n, int f,
                         const float voxel_size[3], const float coors_range[6], int max_points,
                         int max_voxels, float *voxels, int *coors, int coors_stride, int batch_idx0,
                         int *num_points_per_voxel, float *mean, int mean_features, int mean_stride,
                         int *row_count_dev, int rows_capacity, int *overflow_dev,
                         void *const *workspaces, size_t workspace_bytes, hipStream_t st,
                         const SfVoxFuse *fuse) {
  return hard_voxelize_impl(count, points, n, f, voxel_size, coors_range, max_points, max_voxels, voxels,
                            coors, coors_stride, batch_idx0, num_points_per_voxel, mean,
                            mean_features, mean_stride, row_count_dev, rows_capacity, overflow_dev,
                            nullptr, workspaces, workspace_bytes, st, fuse);
}

void sf_int_vox_table(void *workspace, int n, int max_points, uint32_t **table, size_t *words) {
  VoxWorkspace w = carve(workspace, n, max_points);
  *table = w.kf;
  *words = (size_t)w.cap * 3;   // slot table + bound words
}

extern "C" int sf_hard_voxelize(const float *points, int n, int f, const float voxel_size[3],
                                const float coors_range[6], int max_points, int max_voxels,
                                float *voxels, int *coors, int coors_stride, int batch_idx,
                                int *num_points_per_voxel, float *mean, int mean_features,
                                int mean_stride, int *row_count_dev, int rows_capacity,
                                int *overflow_dev, int *voxel_num_host, void *workspace,
                                size_t workspace_bytes, void *stream) {
  SF_CHECK_ARG(n >= 0, "bad shape n=%d", n);
  return hard_voxelize_impl(1, &points, &n, f, voxel_size, coors_range, max_points, max_voxels, voxels,
                            coors, coors_stride, batch_idx, num_points_per_voxel, mean,
                            mean_features, mean_stride, row_count_dev, rows_capacity, overflow_dev,
                            voxel_num_host, &workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int sf_hard_voxelize_static(const float *points, int n, int f,
                                       const float voxel_size[3], const float coors_range[6],
                                       int max_points, int max_voxels, float *voxels, int *coors,
                                       int coors_stride, int batch_idx, int *num_points_per_voxel,
                                       float *mean, int mean_features, int mean_stride,
                                       int *row_count_dev, int rows_capacity, int *overflow_dev,
                                       void *stream) {
  SF_CHECK_ARG(n >= 0 && max_points >= 1, "bad n=%d / max_points=%d", n, max_points);
  const size_t need = carve(nullptr, n, max_points).bytes;
  if (need > kStaticScratchBytes) {
    sf_set_error("sf_hard_voxelize_static: %d points x max_points %d need %zu bytes of scratch, the "
                 "static buffer holds %zu (use sf_hard_voxelize with a caller workspace)",
                 n, max_points, need, kStaticScratchBytes);
    return SF_ERR_WORKSPACE;
  }
  void *ws = nullptr;
  int rc = static_scratch(&ws);
  if (rc != SF_OK) return rc;
  // every word the kernels read is (re)initialised by vox_fill / written earlier in the same call,
  // so the previous call's contents never leak into this one
  return hard_voxelize_impl(1, &points, &n, f, voxel_size, coors_range, max_points, max_voxels, voxels,
                            coors, coors_stride, batch_idx, num_points_per_voxel, mean,
                            mean_features, mean_stride, row_count_dev, rows_capacity, overflow_dev,
                            nullptr, &ws, kStaticScratchBytes, (hipStream_t)stream);
}

extern "C" int sf_voxel_mean(const float *voxels, const int *num_points_per_voxel, const int *m_dev,
                             int m_bound, int max_points, int f, int num_features, float *out,
                             int out_stride, void *stream) {
  SF_CHECK_ARG(m_bound >= 0 && max_points >= 1 && f >= 1 && num_features >= 1 &&
                   num_features <= f && out_stride >= num_features,
               "bad voxel_mean shape");
  if (m_bound == 0) return SF_OK;
  SF_CHECK_ARG(voxels && num_points_per_voxel && out, "NULL tensor pointer");
  hipLaunchKernelGGL(voxel_mean_kernel, dim3(sf_grid((long long)m_bound * out_stride, kThreads)),
                     dim3(kThreads), 0, (hipStream_t)stream, voxels, num_points_per_voxel, m_dev,
                     m_bound, max_points, f, num_features, out, out_stride);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

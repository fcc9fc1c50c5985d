// Rulebook construction for SubMConv3d / SparseConv3d on gfx950.
//
// The reference (spconv 1.x as vendored under mmdet3d/ops/spconv) keeps a dense int32 grid of
// B*D*H*W cells per layer (spconv_ops.h:60-62, 340 MB at the first nuScenes stage) and emits
// per-offset pair lists.  Here the active set lives in an open-addressing hash table sized to
// the active count, and the rulebook is kept OUTPUT-major ("neighbour table"): for every output
// row and kernel offset the input row or -1.  That is the form the output-stationary
// convolution kernel consumes; sf_export_indice_pairs converts it to the reference's tensors.
#include <string.h>

#include "front_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kItems = 1;                  // inputs per thread in the scans (latency-bound: favour parallelism)
constexpr int kChunk = kThreads * kItems;

struct Geo {
  int ks[3], st[3], pad[3];
  int oshape[3];  // output spatial shape (D,H,W)
  int ishape[3];  // input spatial shape
  int K;
};

__device__ __forceinline__ int load_n(const int *n_dev, int n_bound) {
  int n = n_dev ? *n_dev : n_bound;
  return n < n_bound ? n : n_bound;
}

__device__ __forceinline__ uint32_t lin_key(int b, int z, int y, int x, const int shape[3]) {
  return (((uint32_t)b * (uint32_t)shape[0] + (uint32_t)z) * (uint32_t)shape[1] + (uint32_t)y) *
             (uint32_t)shape[2] +
         (uint32_t)x;
}

__global__ void table_fill_kernel(uint32_t *__restrict__ keys, int *__restrict__ vals, int cap,
                                  int *__restrict__ extra, int n_extra, int extra_val) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  for (int i = tid; i < cap; i += stride) {
    keys[i] = SF_EMPTY_KEY;
    vals[i] = SF_INT_MAX;
  }
  for (int i = tid; i < n_extra; i += stride) extra[i] = extra_val;
}

__global__ void hash_build_kernel(const int *__restrict__ indices, const int *n_dev, int n_bound,
                                  Geo g, uint32_t *__restrict__ keys, int *__restrict__ vals,
                                  uint32_t mask, int *overflow_dev) {
  const int n = load_n(n_dev, n_bound);
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    int4 c = reinterpret_cast<const int4 *>(indices)[j];
    int slot = sf_hash_insert(keys, mask, lin_key(c.x, c.y, c.z, c.w, g.ishape));
    if (slot < 0) {
      if (overflow_dev) *overflow_dev = 1;
      continue;
    }
    atomicMin(&vals[slot], j);  // first occurrence wins for duplicate coordinates
  }
}

// Enumerates the outputs reached from one input coordinate in the order of getValidOutPos
// (geometry.h:57-83): ascending kernel-offset index; calls fn(k, oz, oy, ox).
// An output o along a dim is reached through offset kd iff o*stride - pad + kd == in.
template <typename F>
__device__ __forceinline__ void for_each_output(const int4 c, const Geo &g, F fn) {
  for (int k0 = 0; k0 < g.ks[0]; ++k0) {
    int t0 = c.y + g.pad[0] - k0;
    if (t0 < 0 || t0 % g.st[0]) continue;
    int o0 = t0 / g.st[0];
    if (o0 >= g.oshape[0]) continue;
    for (int k1 = 0; k1 < g.ks[1]; ++k1) {
      int t1 = c.z + g.pad[1] - k1;
      if (t1 < 0 || t1 % g.st[1]) continue;
      int o1 = t1 / g.st[1];
      if (o1 >= g.oshape[1]) continue;
      for (int k2 = 0; k2 < g.ks[2]; ++k2) {
        int t2 = c.w + g.pad[2] - k2;
        if (t2 < 0 || t2 % g.st[2]) continue;
        int o2 = t2 / g.st[2];
        if (o2 >= g.oshape[2]) continue;
        fn((k0 * g.ks[1] + k1) * g.ks[2] + k2, o0, o1, o2);
      }
    }
  }
}

// pass 1: insert every reachable output coordinate, remember its first toucher (input*K + k)
__global__ void outset_insert_kernel(const int *__restrict__ indices, const int *n_dev,
                                     int n_bound, Geo g, uint32_t *__restrict__ keys,
                                     uint32_t mask, int *__restrict__ first, int *overflow_dev) {
  const int n = load_n(n_dev, n_bound);
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    int4 c = reinterpret_cast<const int4 *>(indices)[j];
    for_each_output(c, g, [&](int k, int oz, int oy, int ox) {
      int slot = sf_hash_insert(keys, mask, lin_key(c.x, oz, oy, ox, g.oshape));
      if (slot < 0) {
        if (overflow_dev) *overflow_dev = 1;
        return;
      }
      atomicMin(&first[slot], j * g.K + k);
    });
  }
}

// pass 2: per block, how many outputs are first touched by its inputs
__global__ void __launch_bounds__(kThreads)
outset_count_kernel(const int *__restrict__ indices, const int *n_dev, int n_bound, Geo g,
                    const uint32_t *__restrict__ keys, uint32_t mask,
                    const int *__restrict__ first, int *__restrict__ bsum,
                    unsigned *__restrict__ ftmask) {
  __shared__ int smem[17];
  const int n = load_n(n_dev, n_bound);
  if (blockIdx.x * kChunk >= n) return;  // grid is sized from the bound; bsum beyond n is unused
  int c = 0;
  for (int e = 0; e < kItems; ++e) {
    int j = blockIdx.x * kChunk + threadIdx.x * kItems + e;
    if (j >= n) break;
    int4 ci = reinterpret_cast<const int4 *>(indices)[j];
    unsigned fm = 0;  // bit k set: this input is the first toucher of its output under offset k
    for_each_output(ci, g, [&](int k, int oz, int oy, int ox) {
      int slot = sf_hash_find(keys, mask, lin_key(ci.x, oz, oy, ox, g.oshape));
      if (slot >= 0 && first[slot] == j * g.K + k) {
        ++c;
        fm |= 1u << k;
      }
    });
    ftmask[j] = fm;
  }
  int total = sf_block_sum(c, smem);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// pass 3: number the outputs in first-touch order, write outids and the row into the table
__global__ void __launch_bounds__(kThreads)
outset_assign_kernel(const int *__restrict__ indices, const int *n_dev, int n_bound, Geo g,
                     const uint32_t *__restrict__ keys, uint32_t mask,
                     const unsigned *__restrict__ ftmask, const int *__restrict__ bsum, int nblk,
                     int *__restrict__ vals, int *__restrict__ outids, int outids_rows,
                     int *__restrict__ n_out_dev, int *overflow_dev) {
  __shared__ int smem[17];
  const int n = load_n(n_dev, n_bound);
  if (blockIdx.x > 0 && blockIdx.x * kChunk >= n) return;
  const int nblk_used = (n + kChunk - 1) / kChunk;  // only these blocks wrote bsum
  int before, all;
  sf_block_prefix_of(bsum, nblk_used < nblk ? nblk_used : nblk, blockIdx.x, smem, &before, &all);
  int c = 0;
  for (int e = 0; e < kItems; ++e) {
    int j = blockIdx.x * kChunk + threadIdx.x * kItems + e;
    if (j >= n) break;
    c += __popc(ftmask[j]);
  }
  int total;
  int row = before + sf_block_excl_scan(c, smem, &total);
  for (int e = 0; e < kItems; ++e) {
    int j = blockIdx.x * kChunk + threadIdx.x * kItems + e;
    if (j >= n) break;
    const unsigned fm = ftmask[j];
    if (fm == 0) continue;
    int4 ci = reinterpret_cast<const int4 *>(indices)[j];
    for_each_output(ci, g, [&](int k, int oz, int oy, int ox) {
      if ((fm >> k) & 1u) {
        int slot = sf_hash_find(keys, mask, lin_key(ci.x, oz, oy, ox, g.oshape));
        if (slot < 0) return;
        if (row < outids_rows) {
          vals[slot] = row;
          reinterpret_cast<int4 *>(outids)[row] = make_int4(ci.x, oz, oy, ox);
        } else {
          vals[slot] = -1;
          if (overflow_dev) *overflow_dev = 1;
        }
        ++row;
      }
    });
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_out_dev = all < outids_rows ? all : outids_rows;
}

// grid: x over output rows, y = kernel offset (uniform per block: its geometry is scalar work)
__global__ void neighbor_table_kernel(const int *__restrict__ outids, const int *n_out_dev,
                                      int n_out_bound, Geo g, int batch,
                                      const uint32_t *__restrict__ keys,
                                      const int *__restrict__ vals, uint32_t mask,
                                      int identity_center, int *__restrict__ nbr, int nbr_stride) {
  const int n = load_n(n_out_dev, n_out_bound);
  const int k = blockIdx.y;
  const int k2 = k % g.ks[2], k01 = k / g.ks[2];
  const int k1 = k01 % g.ks[1], k0 = k01 / g.ks[1];
  const int d0 = k0 - g.pad[0], d1 = k1 - g.pad[1], d2 = k2 - g.pad[2];
  const bool ident = identity_center && k == (g.K - 1) / 2;
  int *__restrict__ col = nbr + (size_t)k * nbr_stride;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n; o += gridDim.x * blockDim.x) {
    int res = -1;
    if (ident) {
      res = o;
    } else {
      int4 c = reinterpret_cast<const int4 *>(outids)[o];
      int iz = c.y * g.st[0] + d0, iy = c.z * g.st[1] + d1, ix = c.w * g.st[2] + d2;
      if (iz >= 0 && iz < g.ishape[0] && iy >= 0 && iy < g.ishape[1] && ix >= 0 &&
          ix < g.ishape[2] && c.x >= 0 && c.x < batch)
        res = sf_hash_lookup(keys, vals, mask, lin_key(c.x, iz, iy, ix, g.ishape));
    }
    col[o] = res;
  }
}

// ---- fast paths ------------------------------------------------------------------------
// The lookups above are latency bound (one dependent L2 round trip per probe).  The two kernels
// below cut the number of probes instead of trying to speed each one up.

// columns [k0, k1) of the table := -1 for the rows in use
__global__ void nbr_fill_kernel(int *__restrict__ nbr, int nbr_stride, int k0, int k1,
                                const int *n_dev, int n_bound) {
  const int n = load_n(n_dev, n_bound);
  const int k = k0 + blockIdx.y;
  if (k >= k1) return;
  int *__restrict__ col = nbr + (size_t)k * nbr_stride;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n; o += gridDim.x * blockDim.x) col[o] = -1;
}

// the same for two tables on the same rows in one launch (blockIdx.y runs over a's columns, then b's)
__global__ void nbr_fill2_kernel(SfNbrPrefill a, SfNbrPrefill b, const int *n_dev, int n_bound) {
  const int n = load_n(n_dev, n_bound);
  const int ya = a.k1 - a.k0;
  const bool second = (int)blockIdx.y >= ya;
  const SfNbrPrefill &t = second ? b : a;
  const int k = t.k0 + (int)blockIdx.y - (second ? ya : 0);
  if (k >= t.k1) return;
  int *__restrict__ col = t.nbr + (size_t)k * t.stride;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n; o += gridDim.x * blockDim.x) col[o] = -1;
}

// hash_build_kernel + the fill of a table on the same rows (the set's SubM table) in one launch
__global__ void hash_build_fill_kernel(const int *__restrict__ indices, const int *n_dev, int n_bound,
                                       Geo g, uint32_t *__restrict__ keys, int *__restrict__ vals,
                                       uint32_t mask, int *overflow_dev, SfNbrPrefill pf) {
  const int n = load_n(n_dev, n_bound);
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    int4 c = reinterpret_cast<const int4 *>(indices)[j];
    for (int k = pf.k0; k < pf.k1; ++k) pf.nbr[(size_t)k * pf.stride + j] = -1;
    int slot = sf_hash_insert(keys, mask, lin_key(c.x, c.y, c.z, c.w, g.ishape));
    if (slot < 0) {
      if (overflow_dev) *overflow_dev = 1;
      continue;
    }
    atomicMin(&vals[slot], j);  // first occurrence wins for duplicate coordinates
  }
}

// SubMConv3d 3x3x3: if row o finds row i under offset k, then row i finds row o under offset
// 26-k.  Each row looks up the 13 offsets below the centre only, writes its own entry and scatters
// the mirrored one (columns 14..26 are pre-filled with -1; atomicMin on the unsigned view keeps
// the first occurrence among duplicate coordinates).
// The 13 offsets are four x-triples (dz,dy) = (-1,-1) (-1,0) (-1,1) (0,-1) with dx = -1,0,1 and the
// single (0,0,-1).  The three keys of a triple are consecutive integers, i.e. they sit in one or two
// GROUPS of the coordinate hash (sf_group_slot): the kernel loads whole groups -- 16 bytes of keys,
// one or two loads per triple, 7 on average per row instead of 13 probes -- and only the hits load a
// value.  This is a random-access kernel: its floor is the number of L2 requests, not their bytes.
// First-probe collisions (another group's key at the position) continue by single slots, all
// pending offsets of a row together per round.
__device__ __forceinline__ void
subm27_body(int block, int nblocks, const int *__restrict__ outids, const int *n_dev, int n_bound,
            const Geo &g, int batch, const uint32_t *__restrict__ keys, const int *__restrict__ vals,
            uint32_t mask, int *__restrict__ nbr, int nbr_stride) {
  const int n = load_n(n_dev, n_bound);
  const uint4 *__restrict__ keys4 = reinterpret_cast<const uint4 *>(keys);
  const uint32_t gmask = mask >> 2;
  for (int o = block * blockDim.x + threadIdx.x; o < n; o += nblocks * blockDim.x) {
    const int4 c = reinterpret_cast<const int4 *>(outids)[o];
    nbr[(size_t)13 * nbr_stride + o] = o;  // centre: every row pairs with itself
    const bool okb = c.x >= 0 && c.x < batch;
    const bool okl = c.w >= 1, okh = c.w + 1 < g.ishape[2];
    uint32_t key[13], slot[13], kc[13];
    bool ok[13];
    // ---- group loads: all of them in flight before the first compare ------------------------
    uint4 KA[5], KB[4];
    uint32_t hA[5], hB[4];
    bool two[4];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int dz = t < 3 ? -1 : 0, dy = t < 3 ? t - 1 : (t == 3 ? -1 : 0);
      const int iz = c.y + dz, iy = c.z + dy;
      const bool okr = okb && iz >= 0 && iz < g.ishape[0] && iy >= 0 && iy < g.ishape[1];
      const uint32_t kcen = lin_key(c.x, iz, iy, c.w, g.ishape);
      if (t < 4) {
        const uint32_t lo = okl ? kcen - 1u : kcen, hi = okh ? kcen + 1u : kcen;
        hA[t] = sf_hash32(lo >> 2);
        hB[t] = sf_hash32(hi >> 2);
        two[t] = okr && (hi >> 2) != (lo >> 2);
        KA[t] = okr ? keys4[hA[t] & gmask] : make_uint4(SF_EMPTY_KEY, SF_EMPTY_KEY, SF_EMPTY_KEY, SF_EMPTY_KEY);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const int k = 3 * t + d;
          key[k] = kcen + (uint32_t)d - 1u;
          ok[k] = okr && (d == 0 ? okl : (d == 2 ? okh : true));
        }
      } else {
        key[12] = kcen - 1u;
        ok[12] = okr && okl;
        hA[4] = sf_hash32(key[12] >> 2);
        KA[4] = ok[12] ? keys4[hA[4] & gmask] : make_uint4(SF_EMPTY_KEY, SF_EMPTY_KEY, SF_EMPTY_KEY, SF_EMPTY_KEY);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) KB[t] = two[t] ? keys4[hB[t] & gmask] : KA[t];
    // ---- first probe of every offset from the loaded groups ---------------------------------
    auto pick = [](const uint4 &K, uint32_t p) {
      return p == 0 ? K.x : (p == 1 ? K.y : (p == 2 ? K.z : K.w));
    };
#pragma unroll
    for (int k = 0; k < 13; ++k) {
      const int t = k < 12 ? k / 3 : 4;
      bool inB = false;
      if (t < 4) {
        const uint32_t lo = okl ? key[3 * t + 1] - 1u : key[3 * t + 1];
        inB = (key[k] >> 2) != (lo >> 2);
      }
      const uint32_t h = inB ? hB[t < 4 ? t : 0] : hA[t];
      slot[k] = sf_group_slot(h, key[k], mask);
      kc[k] = pick(inB ? KB[t < 4 ? t : 0] : KA[t], slot[k] & 3u);
    }
    int res[13], vc[13];
    bool pend[13], hit[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) {
      hit[k] = ok[k] && kc[k] == key[k];
      pend[k] = ok[k] && !hit[k] && kc[k] != SF_EMPTY_KEY;
      if (pend[k]) slot[k] = sf_next_slot(slot[k], mask);
    }
    // one batch: the values of the hits and the next probe of the collisions
#pragma unroll
    for (int k = 0; k < 13; ++k)
      if (hit[k] || pend[k]) vc[k] = vals[slot[k]];
#pragma unroll
    for (int k = 0; k < 13; ++k)
      if (pend[k]) kc[k] = keys[slot[k]];
#pragma unroll
    for (int k = 0; k < 13; ++k) res[k] = hit[k] ? vc[k] : -1;
    for (uint32_t round = 0; round <= gmask; ++round) {
      bool more = false;
#pragma unroll
      for (int k = 0; k < 13; ++k) {
        if (pend[k]) {
          if (kc[k] == key[k]) {
            res[k] = vc[k];
            pend[k] = false;
          } else if (kc[k] == SF_EMPTY_KEY) {
            pend[k] = false;
          } else {
            slot[k] = sf_next_slot(slot[k], mask);
            more = true;
          }
        }
      }
      if (!more) break;
#pragma unroll
      for (int k = 0; k < 13; ++k) {
        if (pend[k]) {
          kc[k] = keys[slot[k]];
          vc[k] = vals[slot[k]];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 13; ++k) {
      nbr[(size_t)k * nbr_stride + o] = res[k];
      if (res[k] >= 0 && res[k] < n)
        atomicMin(reinterpret_cast<unsigned *>(nbr) + (size_t)(26 - k) * nbr_stride + res[k],
                  (unsigned)o);
    }
  }
}

__global__ void __launch_bounds__(kThreads)
subm27_table_kernel(const int *__restrict__ outids, const int *n_dev, int n_bound, Geo g, int batch,
                    const uint32_t *__restrict__ keys, const int *__restrict__ vals, uint32_t mask,
                    int *__restrict__ nbr, int nbr_stride) {
  subm27_body(blockIdx.x, gridDim.x, outids, n_dev, n_bound, g, batch, keys, vals, mask, nbr,
              nbr_stride);
}

// Strided rulebook, input side.  Every input remembers the table slots of the outputs it reaches
// (slots[e*n_bound + j], e = ordinal in for_each_output order), so the output table is probed
// ONCE per (input, output) -- the count / assign passes and the neighbour table reuse the slots.
__global__ void outset_insert_slots_kernel(const int *__restrict__ indices, const int *n_dev,
                                           int n_bound, Geo g, uint32_t *__restrict__ keys,
                                           uint32_t mask, int *__restrict__ first,
                                           int *__restrict__ slots, int *overflow_dev) {
  const int n = load_n(n_dev, n_bound);
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    int4 c = reinterpret_cast<const int4 *>(indices)[j];
    int e = 0;
    for_each_output(c, g, [&](int k, int oz, int oy, int ox) {
      int slot = sf_hash_insert(keys, mask, lin_key(c.x, oz, oy, ox, g.oshape));
      if (slot < 0) {
        if (overflow_dev) *overflow_dev = 1;
      } else {
        atomicMin(&first[slot], j * g.K + k);
      }
      slots[(size_t)e * n_bound + j] = slot;
      ++e;
    });
  }
}

__global__ void __launch_bounds__(kThreads)
outset_count_slots_kernel(const int *__restrict__ indices, const int *n_dev, int n_bound, Geo g,
                          const int *__restrict__ first, const int *__restrict__ slots,
                          int *__restrict__ bsum, unsigned *__restrict__ ftmask) {
  __shared__ int smem[17];
  const int n = load_n(n_dev, n_bound);
  if (blockIdx.x * kChunk >= n) return;
  int c = 0;
  const int j = blockIdx.x * kChunk + threadIdx.x;
  if (j < n) {
    int4 ci = reinterpret_cast<const int4 *>(indices)[j];
    unsigned fm = 0;
    int e = 0;
    for_each_output(ci, g, [&](int k, int, int, int) {
      const int slot = slots[(size_t)e * n_bound + j];
      ++e;
      if (slot >= 0 && first[slot] == j * g.K + k) {
        ++c;
        fm |= 1u << k;
      }
    });
    ftmask[j] = fm;
  }
  int total = sf_block_sum(c, smem);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kThreads)
outset_assign_slots_kernel(const int *__restrict__ indices, const int *n_dev, int n_bound, Geo g,
                           const int *__restrict__ slots, const unsigned *__restrict__ ftmask,
                           const int *__restrict__ bsum, int nblk, int *__restrict__ vals,
                           int *__restrict__ outids, int outids_rows, int *__restrict__ n_out_dev,
                           int *overflow_dev) {
  __shared__ int smem[17];
  const int n = load_n(n_dev, n_bound);
  if (blockIdx.x > 0 && blockIdx.x * kChunk >= n) return;
  const int nblk_used = (n + kChunk - 1) / kChunk;
  int before, all;
  sf_block_prefix_of(bsum, nblk_used < nblk ? nblk_used : nblk, blockIdx.x, smem, &before, &all);
  const int j = blockIdx.x * kChunk + threadIdx.x;
  const unsigned fm = j < n ? ftmask[j] : 0u;
  int total;
  int row = before + sf_block_excl_scan(__popc(fm), smem, &total);
  if (fm != 0) {
    int4 ci = reinterpret_cast<const int4 *>(indices)[j];
    int e = 0;
    for_each_output(ci, g, [&](int k, int oz, int oy, int ox) {
      const int slot = slots[(size_t)e * n_bound + j];
      ++e;
      if (((fm >> k) & 1u) && slot >= 0) {
        if (row < outids_rows) {
          vals[slot] = row;
          reinterpret_cast<int4 *>(outids)[row] = make_int4(ci.x, oz, oy, ox);
        } else {
          vals[slot] = -1;
          if (overflow_dev) *overflow_dev = 1;
        }
        ++row;
      }
    });
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_out_dev = all < outids_rows ? all : outids_rows;
}

// nbr[k][output row] = input row, scattered from the input side (the table is pre-filled with
// -1; atomicMin on the unsigned view: first occurrence wins among duplicate input coordinates)
__global__ void outset_scatter_table_kernel(const int *__restrict__ indices, const int *n_dev,
                                            int n_bound, Geo g, const int *__restrict__ slots,
                                            const int *__restrict__ vals, int *__restrict__ nbr,
                                            int nbr_stride, int nbr_rows) {
  const int n = load_n(n_dev, n_bound);
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    int4 c = reinterpret_cast<const int4 *>(indices)[j];
    int e = 0;
    for_each_output(c, g, [&](int k, int, int, int) {
      const int slot = slots[(size_t)e * n_bound + j];
      ++e;
      if (slot < 0) return;
      const int orow = vals[slot];
      if (orow >= 0 && orow < nbr_rows)
        atomicMin(reinterpret_cast<unsigned *>(nbr) + (size_t)k * nbr_stride + orow, (unsigned)j);
    });
  }
}

// Kernel extents <= 3 per dimension (every conv of SparseEncoder): the <= 27 (offset, output)
// candidates of one input are enumerated branch-free into registers, so that the table accesses
// of ALL of them are in flight together.  Order = ascending offset index = for_each_output order;
// ord[ci] is the candidate's ordinal among the valid ones (its row in the slots array).
__device__ __forceinline__ void enumerate27(const int4 c, const Geo &g, bool (&ok)[27],
                                            int (&kidx)[27], uint32_t (&key)[27], int (&ord)[27]) {
  bool v[3][3];
  int od[3][3];
  const int in[3] = {c.y, c.z, c.w};
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int t = in[d] + g.pad[d] - kd;
      const int o = t / g.st[d];
      v[d][kd] = kd < g.ks[d] && t >= 0 && o * g.st[d] == t && o < g.oshape[d];
      od[d][kd] = o;
    }
  int e = 0;
#pragma unroll
  for (int k0 = 0; k0 < 3; ++k0)
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
      for (int k2 = 0; k2 < 3; ++k2) {
        const int ci = (k0 * 3 + k1) * 3 + k2;
        ok[ci] = v[0][k0] && v[1][k1] && v[2][k2];
        kidx[ci] = (k0 * g.ks[1] + k1) * g.ks[2] + k2;
        key[ci] = lin_key(c.x, od[0][k0], od[1][k1], od[2][k2], g.oshape);
        ord[ci] = e;
        e += ok[ci] ? 1 : 0;
      }
}

__global__ void __launch_bounds__(kThreads)
outset_count27_kernel(const int *__restrict__ indices, const int *n_dev, int n_bound, Geo g,
                      const int *__restrict__ first, const int *__restrict__ slots,
                      int *__restrict__ bsum, unsigned *__restrict__ ftmask) {
  __shared__ int smem[17];
  const int n = load_n(n_dev, n_bound);
  if (blockIdx.x * kChunk >= n) return;
  int c = 0;
  const int j = blockIdx.x * kChunk + threadIdx.x;
  if (j < n) {
    const int4 ci = reinterpret_cast<const int4 *>(indices)[j];
    bool ok[27];
    int kidx[27], ord[27], sl[27], fv[27];
    uint32_t key[27];
    enumerate27(ci, g, ok, kidx, key, ord);
#pragma unroll
    for (int i = 0; i < 27; ++i) sl[i] = ok[i] ? slots[(size_t)ord[i] * n_bound + j] : -1;
#pragma unroll
    for (int i = 0; i < 27; ++i) fv[i] = sl[i] >= 0 ? first[sl[i]] : -1;
    unsigned fm = 0;
#pragma unroll
    for (int i = 0; i < 27; ++i)
      if (sl[i] >= 0 && fv[i] == j * g.K + kidx[i]) {
        ++c;
        fm |= 1u << kidx[i];
      }
    ftmask[j] = fm;
  }
  int total = sf_block_sum(c, smem);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

__device__ __forceinline__ void
scatter27_body(int block, int nblocks, const int *__restrict__ indices, const int *n_dev, int n_bound,
               const Geo &g, const int *__restrict__ slots, const int *__restrict__ vals,
               int *__restrict__ nbr, int nbr_stride, int nbr_rows) {
  const int n = load_n(n_dev, n_bound);
  for (int j = block * blockDim.x + threadIdx.x; j < n; j += nblocks * blockDim.x) {
    const int4 c = reinterpret_cast<const int4 *>(indices)[j];
    bool ok[27];
    int kidx[27], ord[27], sl[27], orow[27];
    uint32_t key[27];
    enumerate27(c, g, ok, kidx, key, ord);
#pragma unroll
    for (int i = 0; i < 27; ++i) sl[i] = ok[i] ? slots[(size_t)ord[i] * n_bound + j] : -1;
#pragma unroll
    for (int i = 0; i < 27; ++i) orow[i] = sl[i] >= 0 ? vals[sl[i]] : -1;
#pragma unroll
    for (int i = 0; i < 27; ++i)
      if (orow[i] >= 0 && orow[i] < nbr_rows)
        atomicMin(reinterpret_cast<unsigned *>(nbr) + (size_t)kidx[i] * nbr_stride + orow[i],
                  (unsigned)j);
  }
}

__global__ void __launch_bounds__(kThreads)
outset_scatter27_kernel(const int *__restrict__ indices, const int *n_dev, int n_bound, Geo g,
                        const int *__restrict__ slots, const int *__restrict__ vals,
                        int *__restrict__ nbr, int nbr_stride, int nbr_rows) {
  scatter27_body(blockIdx.x, gridDim.x, indices, n_dev, n_bound, g, slots, vals, nbr, nbr_stride,
                 nbr_rows);
}

// The table scatter of a strided rulebook and the SubM table of its OUTPUT set depend on the same
// predecessor (the row assignment) and on nothing of each other: one launch, the first
// `scatter_blocks` workgroups scatter, the rest look neighbours up -- two latency-bound kernels
// side by side instead of one behind the other.
struct ScatterArgs {
  const int *indices, *n_dev;
  int n_bound;
  Geo g;
  const int *slots, *vals;
  int *nbr;
  int nbr_stride, nbr_rows;
};
__global__ void __launch_bounds__(kThreads)
scatter27_subm27_kernel(ScatterArgs sa, int scatter_blocks, const int *__restrict__ outids,
                        const int *n_dev, int n_bound, Geo g, int batch,
                        const uint32_t *__restrict__ keys, const int *__restrict__ vals, uint32_t mask,
                        int *__restrict__ nbr, int nbr_stride) {
  if ((int)blockIdx.x < scatter_blocks)
    scatter27_body(blockIdx.x, scatter_blocks, sa.indices, sa.n_dev, sa.n_bound, sa.g, sa.slots, sa.vals,
                   sa.nbr, sa.nbr_stride, sa.nbr_rows);
  else
    subm27_body(blockIdx.x - scatter_blocks, gridDim.x - scatter_blocks, outids, n_dev, n_bound, g,
                batch, keys, vals, mask, nbr, nbr_stride);
}

static_assert(sizeof(Geo) <= sizeof(((SfScatterJob *)0)->geo), "SfScatterJob::geo too small");

// ---- export to the reference's (K,2,L) pair lists ------------------------------------
__global__ void fill_int_kernel(int *__restrict__ p, long long n, int v) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    p[i] = v;
}

__global__ void export_invert_kernel(const int *__restrict__ nbr, int nbr_stride, int K,
                                     const int *n_out_dev, int n_out_bound, int n_in,
                                     int *__restrict__ inv) {
  const int n = load_n(n_out_dev, n_out_bound);
  const long long total = (long long)n * K;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    int k = (int)(e / n), o = (int)(e - (long long)k * n);
    int in = nbr[(size_t)k * nbr_stride + o];
    if (in >= 0 && in < n_in) inv[(size_t)k * n_in + in] = o;
  }
}

__global__ void __launch_bounds__(kThreads)
export_count_kernel(const int *__restrict__ inv, int n_in, int nblk, int *__restrict__ bsum) {
  __shared__ int smem[17];
  const int k = blockIdx.y;
  int c = 0;
  for (int e = 0; e < kItems; ++e) {
    int i = blockIdx.x * kChunk + threadIdx.x * kItems + e;
    if (i < n_in && inv[(size_t)k * n_in + i] >= 0) ++c;
  }
  int total = sf_block_sum(c, smem);
  if (threadIdx.x == 0) bsum[k * nblk + blockIdx.x] = total;
}

__global__ void __launch_bounds__(kThreads)
export_write_kernel(const int *__restrict__ inv, int n_in, int nblk, const int *__restrict__ bsum,
                    int *__restrict__ pairs, int *__restrict__ num) {
  __shared__ int smem[17];
  const int k = blockIdx.y;
  int before, all;
  sf_block_prefix_of(bsum + k * nblk, nblk, blockIdx.x, smem, &before, &all);
  int c = 0;
  int o[kItems];
  for (int e = 0; e < kItems; ++e) {
    int i = blockIdx.x * kChunk + threadIdx.x * kItems + e;
    o[e] = (i < n_in) ? inv[(size_t)k * n_in + i] : -1;
    if (o[e] >= 0) ++c;
  }
  int total;
  int pos = before + sf_block_excl_scan(c, smem, &total);
  for (int e = 0; e < kItems; ++e) {
    if (o[e] < 0) continue;
    int i = blockIdx.x * kChunk + threadIdx.x * kItems + e;
    pairs[((size_t)k * 2 + 0) * n_in + pos] = i;
    pairs[((size_t)k * 2 + 1) * n_in + pos] = o[e];
    ++pos;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) num[k] = all;
}

__global__ void pairs_to_table_kernel(const int *__restrict__ pairs, const int *__restrict__ num,
                                      int K, int L, int *__restrict__ nbr, int nbr_stride,
                                      int n_out) {
  const long long total = (long long)K * L;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    int k = (int)(e / L), i = (int)(e - (long long)k * L);
    if (i >= num[k]) continue;
    int in = pairs[((size_t)k * 2 + 0) * L + i], o = pairs[((size_t)k * 2 + 1) * L + i];
    if (in >= 0 && o >= 0 && o < n_out) nbr[(size_t)k * nbr_stride + o] = in;
  }
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

int make_geo(const int ishape[3], const int oshape[3], const int ks[3], const int st[3],
             const int pad[3], int batch, Geo *g) {
  SF_CHECK_ARG(ks && st && pad, "ksize / stride / padding is NULL");
  g->K = 1;
  for (int i = 0; i < 3; ++i) {
    SF_CHECK_ARG(ks[i] >= 1 && ks[i] <= 7 && st[i] >= 1 && pad[i] >= 0,
                 "unsupported ksize/stride/padding at dim %d: %d/%d/%d", i, ks[i], st[i], pad[i]);
    g->ks[i] = ks[i];
    g->st[i] = st[i];
    g->pad[i] = pad[i];
    g->ishape[i] = ishape ? ishape[i] : 0;
    g->oshape[i] = oshape ? oshape[i] : 0;
    g->K *= ks[i];
  }
  const int *shapes[2] = {ishape, oshape};
  for (int s = 0; s < 2; ++s) {
    if (!shapes[s]) continue;
    long long vol = batch;
    for (int i = 0; i < 3; ++i) {
      SF_CHECK_ARG(shapes[s][i] > 0, "spatial shape must be positive");
      vol *= shapes[s][i];
    }
    if (vol >= 0xFFFFFFFFll) {
      sf_set_error("batch*D*H*W = %lld does not fit 32-bit coordinate keys", vol);
      return SF_ERR_CAPACITY;
    }
  }
  return SF_OK;
}

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" int sf_hash_capacity(long long n_bound) {
  long long want = 2 * (n_bound > 0 ? n_bound : 1);
  long long cap = 1024;
  while (cap < want && cap < (1ll << 30)) cap <<= 1;
  return (int)cap;
}

extern "C" int sf_hash_build(const int *indices, const int *n_dev, int n_bound, const int shape[3],
                             int batch, uint32_t *keys, int *vals, int capacity, int *overflow_dev,
                             void *stream) {
  return sf_int_hash_build(indices, n_dev, n_bound, shape, batch, keys, vals, capacity, overflow_dev, 0,
                           SfNbrPrefill{nullptr, 0, 0, 0}, (hipStream_t)stream);
}

int sf_int_hash_build(const int *indices, const int *n_dev, int n_bound, const int shape[3], int batch,
                      uint32_t *keys, int *vals, int capacity, int *overflow_dev, int keys_ready,
                      SfNbrPrefill pf, hipStream_t st) {
  SF_CHECK_ARG(n_bound >= 0 && batch >= 1 && shape, "bad n_bound/batch/shape");
  SF_CHECK_ARG(keys && vals && is_pow2(capacity) && capacity >= n_bound,
               "hash table NULL or capacity %d not a power of two >= n_bound %d", capacity,
               n_bound);
  SF_CHECK_ARG(n_bound == 0 || indices, "indices is NULL");
  Geo g;
  const int one[3] = {1, 1, 1}, zero[3] = {0, 0, 0};
  int rc = make_geo(shape, nullptr, one, one, zero, batch, &g);
  if (rc != SF_OK) return rc;
  if (!keys_ready) {
    hipLaunchKernelGGL(table_fill_kernel, dim3(sf_grid(capacity, kThreads)), dim3(kThreads), 0, st,
                       keys, vals, capacity, (int *)nullptr, 0, 0);
    SF_CHECK_LAUNCH();
  }
  if (n_bound > 0) {
    if (pf.nbr && pf.k1 > pf.k0)
      hipLaunchKernelGGL(hash_build_fill_kernel, dim3(sf_grid(n_bound, kThreads)), dim3(kThreads), 0, st,
                         indices, n_dev, n_bound, g, keys, vals, (uint32_t)(capacity - 1), overflow_dev,
                         pf);
    else
      hipLaunchKernelGGL(hash_build_kernel, dim3(sf_grid(n_bound, kThreads)), dim3(kThreads), 0, st,
                         indices, n_dev, n_bound, g, keys, vals, (uint32_t)(capacity - 1),
                         overflow_dev);
    SF_CHECK_LAUNCH();
  }
  return SF_OK;
}

extern "C" size_t sf_conv_output_set_workspace(int n_bound, int out_capacity) {
  if (n_bound < 0 || out_capacity < 0) return 0;
  int nblk = sf_div_up(n_bound > 0 ? n_bound : 1, kChunk);
  return align_up((size_t)out_capacity * 4) + align_up((size_t)nblk * 4) +
         align_up((size_t)(n_bound > 0 ? n_bound : 1) * 4);
}

extern "C" int sf_conv_output_set(const int *indices, const int *n_dev, int n_bound, int batch,
                                  const int out_shape[3], const int ksize[3], const int stride[3],
                                  const int padding[3], uint32_t *out_keys, int *out_vals,
                                  int out_capacity, int *outids, int outids_rows, int *n_out_dev,
                                  int *overflow_dev, void *workspace, size_t workspace_bytes,
                                  void *stream) {
  hipStream_t st = (hipStream_t)stream;
  SF_CHECK_ARG(n_bound >= 0 && batch >= 1 && out_shape, "bad n_bound/batch/out_shape");
  SF_CHECK_ARG(out_keys && out_vals && is_pow2(out_capacity), "output table NULL or capacity %d "
               "not a power of two", out_capacity);
  SF_CHECK_ARG(outids && outids_rows >= 0 && n_out_dev, "outids / n_out_dev is NULL");
  SF_CHECK_ARG(n_bound == 0 || indices, "indices is NULL");
  Geo g;
  int rc = make_geo(nullptr, out_shape, ksize, stride, padding, batch, &g);
  if (rc != SF_OK) return rc;
  SF_CHECK_ARG(g.K <= 32, "kernel volume %d > 32: the first-toucher mask is one 32-bit word", g.K);
  SF_CHECK_ARG((long long)n_bound * g.K < 0x7FFFFFFFll, "n_bound*K overflows int32");
  size_t need = sf_conv_output_set_workspace(n_bound, out_capacity);
  if (!workspace || workspace_bytes < need) {
    sf_set_error("workspace too small: %zu < %zu bytes", workspace_bytes, need);
    return SF_ERR_WORKSPACE;
  }
  int *first = (int *)workspace;
  int *bsum = (int *)((char *)workspace + align_up((size_t)out_capacity * 4));
  const int nblk = sf_div_up(n_bound > 0 ? n_bound : 1, kChunk);
  unsigned *ftmask = (unsigned *)((char *)bsum + align_up((size_t)nblk * 4));
  const uint32_t mask = (uint32_t)(out_capacity - 1);
  hipLaunchKernelGGL(table_fill_kernel, dim3(sf_grid(out_capacity, kThreads)), dim3(kThreads), 0,
                     st, out_keys, first, out_capacity, out_vals, out_capacity, -1);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(outset_insert_kernel, dim3(sf_grid(n_bound, kThreads)), dim3(kThreads), 0, st,
                     indices, n_dev, n_bound, g, out_keys, mask, first, overflow_dev);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(outset_count_kernel, dim3(nblk), dim3(kThreads), 0, st, indices, n_dev,
                     n_bound, g, (const uint32_t *)out_keys, mask, (const int *)first, bsum, ftmask);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(outset_assign_kernel, dim3(nblk), dim3(kThreads), 0, st, indices, n_dev,
                     n_bound, g, (const uint32_t *)out_keys, mask, (const unsigned *)ftmask,
                     (const int *)bsum, nblk, out_vals, outids, outids_rows, n_out_dev,
                     overflow_dev);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

static int conv_reach(const int ks[3], const int st[3]) {
  int r = 1;
  for (int i = 0; i < 3; ++i) r *= (ks[i] + st[i] - 1) / st[i];
  return r;
}

extern "C" size_t sf_strided_rulebook_workspace(int n_bound, int out_capacity, const int ksize[3],
                                                const int stride[3]) {
  if (n_bound < 0 || out_capacity < 0 || !ksize || !stride) return 0;
  for (int i = 0; i < 3; ++i)
    if (ksize[i] < 1 || stride[i] < 1) return 0;
  const size_t nb = (size_t)(n_bound > 0 ? n_bound : 1);
  return sf_conv_output_set_workspace(n_bound, out_capacity) +
         align_up(nb * 4 * (size_t)conv_reach(ksize, stride));
}

int sf_int_strided_rulebook(const int *indices, const int *n_dev, int n_bound, int batch,
                            const int out_shape[3], const int ksize[3], const int stride[3],
                            const int padding[3], uint32_t *out_keys, int *out_vals,
                            int out_capacity, int *outids, int outids_rows, int *n_out_dev,
                            int *overflow_dev, int *nbr, int nbr_stride, void *workspace,
                            size_t workspace_bytes, hipStream_t st, const SfStridedFuse *fuse,
                            SfScatterJob *job) {
  SF_CHECK_ARG(n_bound >= 0 && batch >= 1 && out_shape, "bad n_bound/batch/out_shape");
  SF_CHECK_ARG(out_keys && out_vals && is_pow2(out_capacity), "output table NULL or capacity %d "
               "not a power of two", out_capacity);
  SF_CHECK_ARG(outids && outids_rows >= 0 && n_out_dev, "outids / n_out_dev is NULL");
  SF_CHECK_ARG(nbr && nbr_stride >= outids_rows, "nbr NULL or nbr_stride %d < outids_rows %d",
               nbr_stride, outids_rows);
  SF_CHECK_ARG(n_bound == 0 || indices, "indices is NULL");
  Geo g;
  int rc = make_geo(nullptr, out_shape, ksize, stride, padding, batch, &g);
  if (rc != SF_OK) return rc;
  SF_CHECK_ARG(g.K <= 32, "kernel volume %d > 32: the first-toucher mask is one 32-bit word", g.K);
  SF_CHECK_ARG((long long)n_bound * g.K < 0x7FFFFFFFll, "n_bound*K overflows int32");
  const size_t need = sf_strided_rulebook_workspace(n_bound, out_capacity, ksize, stride);
  if (!workspace || workspace_bytes < need) {
    sf_set_error("workspace too small: %zu < %zu bytes", workspace_bytes, need);
    return SF_ERR_WORKSPACE;
  }
  int *first = (fuse && fuse->first) ? fuse->first : (int *)workspace;
  int *bsum = (int *)((char *)workspace + align_up((size_t)out_capacity * 4));
  const int nblk = sf_div_up(n_bound > 0 ? n_bound : 1, kChunk);
  unsigned *ftmask = (unsigned *)((char *)bsum + align_up((size_t)nblk * 4));
  int *slots = (int *)((char *)workspace + sf_conv_output_set_workspace(n_bound, out_capacity));
  const uint32_t mask = (uint32_t)(out_capacity - 1);
  if (!(fuse && fuse->tables_ready)) {
    // (the values need no fill: every slot a later kernel reads was given its row, or -1, by the
    // assign pass)
    hipLaunchKernelGGL(table_fill_kernel, dim3(sf_grid(out_capacity, kThreads)), dim3(kThreads), 0,
                       st, out_keys, first, out_capacity, (int *)nullptr, 0, -1);
    SF_CHECK_LAUNCH();
  }
  const bool small = ksize[0] <= 3 && ksize[1] <= 3 && ksize[2] <= 3;  // batched-probe kernels
  // (a batched-probe insert -- all candidates' reads, then all claims, in rounds -- measured
  // SLOWER than the sequential one, 50 vs 34 us: most inserts find their key already present on
  // the first read, and the batched form pays 27 masked steps per round)
  hipLaunchKernelGGL(outset_insert_slots_kernel, dim3(sf_grid(n_bound, kThreads)), dim3(kThreads),
                     0, st, indices, n_dev, n_bound, g, out_keys, mask, first, slots, overflow_dev);
  SF_CHECK_LAUNCH();
  if (small)
    hipLaunchKernelGGL(outset_count27_kernel, dim3(nblk), dim3(kThreads), 0, st, indices, n_dev,
                       n_bound, g, (const int *)first, (const int *)slots, bsum, ftmask);
  else
    hipLaunchKernelGGL(outset_count_slots_kernel, dim3(nblk), dim3(kThreads), 0, st, indices,
                       n_dev, n_bound, g, (const int *)first, (const int *)slots, bsum, ftmask);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(outset_assign_slots_kernel, dim3(nblk), dim3(kThreads), 0, st, indices, n_dev,
                     n_bound, g, (const int *)slots, (const unsigned *)ftmask, (const int *)bsum,
                     nblk, out_vals, outids, outids_rows, n_out_dev, overflow_dev);
  SF_CHECK_LAUNCH();
  if (outids_rows > 0) {   // -1 for the rows in use: this table and, in the same launch, a later one
    SfNbrPrefill pf_next = {};
    if (fuse && fuse->pf_next.nbr) pf_next = fuse->pf_next;
    int gx = sf_div_up(outids_rows, kThreads);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(nbr_fill2_kernel, dim3(gx, g.K + (pf_next.k1 - pf_next.k0)), dim3(kThreads), 0, st,
                       SfNbrPrefill{nbr, nbr_stride, 0, g.K}, pf_next, (const int *)n_out_dev, outids_rows);
    SF_CHECK_LAUNCH();
  }
  SfScatterJob j = {};
  j.indices = indices;
  j.n_dev = n_dev;
  j.n_bound = n_bound;
  memcpy(j.geo, &g, sizeof(Geo));
  j.slots = slots;
  j.vals = out_vals;
  j.nbr = nbr;
  j.nbr_stride = nbr_stride;
  j.nbr_rows = outids_rows;
  j.small = small ? 1 : 0;
  if (fuse && fuse->defer_scatter && job) {
    *job = j;
    return SF_OK;
  }
  return sf_int_scatter(&j, st);
}

int sf_int_scatter(const SfScatterJob *job, hipStream_t st) {
  if (job->nbr_rows <= 0 || job->n_bound <= 0) return SF_OK;
  Geo g;
  memcpy(&g, job->geo, sizeof(Geo));
  if (job->small)
    hipLaunchKernelGGL(outset_scatter27_kernel, dim3(sf_grid(job->n_bound, kThreads)), dim3(kThreads),
                       0, st, job->indices, job->n_dev, job->n_bound, g, job->slots, job->vals,
                       job->nbr, job->nbr_stride, job->nbr_rows);
  else
    hipLaunchKernelGGL(outset_scatter_table_kernel, dim3(sf_grid(job->n_bound, kThreads)),
                       dim3(kThreads), 0, st, job->indices, job->n_dev, job->n_bound, g, job->slots,
                       job->vals, job->nbr, job->nbr_stride, job->nbr_rows);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_strided_rulebook(const int *indices, const int *n_dev, int n_bound, int batch,
                                   const int out_shape[3], const int ksize[3], const int stride[3],
                                   const int padding[3], uint32_t *out_keys, int *out_vals,
                                   int out_capacity, int *outids, int outids_rows, int *n_out_dev,
                                   int *overflow_dev, int *nbr, int nbr_stride, void *workspace,
                                   size_t workspace_bytes, void *stream) {
  return sf_int_strided_rulebook(indices, n_dev, n_bound, batch, out_shape, ksize, stride, padding,
                                 out_keys, out_vals, out_capacity, outids, outids_rows, n_out_dev,
                                 overflow_dev, nbr, nbr_stride, workspace, workspace_bytes,
                                 (hipStream_t)stream, nullptr, nullptr);
}

int sf_int_subm27(const int *outids, const int *n_dev, int n_bound, int batch, const int in_shape[3],
                  const uint32_t *keys, const int *vals, int capacity, int *nbr, int nbr_stride,
                  int prefilled, const SfScatterJob *job, hipStream_t st) {
  Geo g;
  const int three[3] = {3, 3, 3}, one[3] = {1, 1, 1};
  int rc = make_geo(in_shape, nullptr, three, one, one, batch, &g);
  if (rc != SF_OK) return rc;
  int gx = sf_div_up(n_bound, kThreads);
  if (gx > 4096) gx = 4096;
  if (gx < 1) gx = 1;
  if (!prefilled) {
    hipLaunchKernelGGL(nbr_fill_kernel, dim3(gx > 1024 ? 1024 : gx, 13), dim3(kThreads), 0, st, nbr,
                       nbr_stride, 14, 27, n_dev, n_bound);
    SF_CHECK_LAUNCH();
  }
  const uint32_t mask = (uint32_t)(capacity - 1);
  if (job && job->small && job->nbr_rows > 0 && job->n_bound > 0) {
    ScatterArgs sa;
    sa.indices = job->indices;
    sa.n_dev = job->n_dev;
    sa.n_bound = job->n_bound;
    memcpy(&sa.g, job->geo, sizeof(Geo));
    sa.slots = job->slots;
    sa.vals = job->vals;
    sa.nbr = job->nbr;
    sa.nbr_stride = job->nbr_stride;
    sa.nbr_rows = job->nbr_rows;
    const int sb = sf_grid(job->n_bound, kThreads);
    hipLaunchKernelGGL(scatter27_subm27_kernel, dim3(sb + gx), dim3(kThreads), 0, st, sa, sb, outids,
                       n_dev, n_bound, g, batch, keys, vals, mask, nbr, nbr_stride);
    SF_CHECK_LAUNCH();
    return SF_OK;
  }
  if (job) {
    rc = sf_int_scatter(job, st);
    if (rc != SF_OK) return rc;
  }
  hipLaunchKernelGGL(subm27_table_kernel, dim3(gx), dim3(kThreads), 0, st, outids, n_dev, n_bound, g,
                     batch, keys, vals, mask, nbr, nbr_stride);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_neighbor_table(const int *outids, const int *n_out_dev, int n_out_bound,
                                 int batch, const int in_shape[3], const int ksize[3],
                                 const int stride[3], const int padding[3],
                                 const uint32_t *in_keys, const int *in_vals, int in_capacity,
                                 int identity_center, int *nbr, int nbr_stride, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  SF_CHECK_ARG(n_out_bound >= 0 && batch >= 1 && in_shape, "bad n_out_bound/batch/in_shape");
  SF_CHECK_ARG(in_keys && in_vals && is_pow2(in_capacity), "input table NULL or capacity %d not "
               "a power of two", in_capacity);
  SF_CHECK_ARG(nbr && nbr_stride >= n_out_bound, "nbr NULL or nbr_stride %d < n_out_bound %d",
               nbr_stride, n_out_bound);
  Geo g;
  int rc = make_geo(in_shape, nullptr, ksize, stride, padding, batch, &g);
  if (rc != SF_OK) return rc;
  SF_CHECK_ARG(!identity_center || (g.K % 2 == 1), "identity_center needs an odd kernel volume");
  if (n_out_bound == 0) return SF_OK;
  SF_CHECK_ARG(outids, "outids is NULL");
  if (identity_center && g.K == 27 && ksize[0] == 3 && ksize[1] == 3 && stride[0] == 1 &&
      stride[1] == 1 && stride[2] == 1 && padding[0] == 1 && padding[1] == 1 && padding[2] == 1) {
    // SubMConv3d 3x3x3: half the probes, the other half mirrored (subm27_table_kernel)
    return sf_int_subm27(outids, n_out_dev, n_out_bound, batch, in_shape, in_keys, in_vals,
                         in_capacity, nbr, nbr_stride, 0, nullptr, st);
  }
  // lookup chains are latency-bound: one (offset, row) per thread where the bound allows it
  int nt_x = sf_div_up(n_out_bound, kThreads);
  if (nt_x > 2048) nt_x = 2048;
  hipLaunchKernelGGL(neighbor_table_kernel, dim3(nt_x, g.K), dim3(kThreads), 0, st, outids,
                     n_out_dev, n_out_bound, g, batch, in_keys, in_vals,
                     (uint32_t)(in_capacity - 1), identity_center, nbr, nbr_stride);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" size_t sf_export_indice_pairs_workspace(int K, int n_in) {
  if (K < 1 || n_in < 0) return 0;
  int nblk = sf_div_up(n_in > 0 ? n_in : 1, kChunk);
  return align_up((size_t)K * (size_t)(n_in > 0 ? n_in : 1) * 4) + align_up((size_t)K * nblk * 4);
}

extern "C" int sf_export_indice_pairs(const int *nbr, int nbr_stride, int K, const int *n_out_dev,
                                      int n_out_bound, int n_in, int *indice_pairs,
                                      int *indice_num, void *workspace, size_t workspace_bytes,
                                      void *stream) {
  hipStream_t st = (hipStream_t)stream;
  SF_CHECK_ARG(K >= 1 && n_in >= 0 && n_out_bound >= 0 && nbr_stride >= n_out_bound,
               "bad K/n_in/n_out_bound/nbr_stride");
  SF_CHECK_ARG(indice_num, "indice_num is NULL");
  size_t need = sf_export_indice_pairs_workspace(K, n_in);
  if (!workspace || workspace_bytes < need) {
    sf_set_error("workspace too small: %zu < %zu bytes", workspace_bytes, need);
    return SF_ERR_WORKSPACE;
  }
  SF_FILL_U32(indice_num, K, 0u, st);
  if (n_in == 0) return SF_OK;
  SF_CHECK_ARG(nbr && indice_pairs, "nbr / indice_pairs is NULL");
  int *inv = (int *)workspace;
  int *bsum = (int *)((char *)workspace + align_up((size_t)K * n_in * 4));
  const int nblk = sf_div_up(n_in, kChunk);
  const long long cells = (long long)K * n_in;
  hipLaunchKernelGGL(fill_int_kernel, dim3(sf_grid(cells, kThreads)), dim3(kThreads), 0, st, inv,
                     cells, -1);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(fill_int_kernel, dim3(sf_grid(2 * cells, kThreads)), dim3(kThreads), 0, st,
                     indice_pairs, 2 * cells, -1);
  SF_CHECK_LAUNCH();
  if (n_out_bound > 0) {
    hipLaunchKernelGGL(export_invert_kernel, dim3(sf_grid((long long)n_out_bound * K, kThreads)),
                       dim3(kThreads), 0, st, nbr, nbr_stride, K, n_out_dev, n_out_bound, n_in,
                       inv);
    SF_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(export_count_kernel, dim3(nblk, K), dim3(kThreads), 0, st, (const int *)inv,
                     n_in, nblk, bsum);
  SF_CHECK_LAUNCH();
  hipLaunchKernelGGL(export_write_kernel, dim3(nblk, K), dim3(kThreads), 0, st, (const int *)inv,
                     n_in, nblk, (const int *)bsum, indice_pairs, indice_num);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_indice_pairs_to_table(const int *indice_pairs, const int *indice_num, int K,
                                        int L, int n_out, int *nbr, int nbr_stride, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  SF_CHECK_ARG(K >= 1 && L >= 0 && n_out >= 0 && nbr_stride >= n_out, "bad K/L/n_out/nbr_stride");
  if (n_out == 0) return SF_OK;
  SF_CHECK_ARG(nbr && indice_num, "nbr / indice_num is NULL");
  const long long cells = (long long)K * nbr_stride;
  hipLaunchKernelGGL(fill_int_kernel, dim3(sf_grid(cells, kThreads)), dim3(kThreads), 0, st, nbr,
                     cells, -1);
  SF_CHECK_LAUNCH();
  if (L == 0) return SF_OK;
  SF_CHECK_ARG(indice_pairs, "indice_pairs is NULL");
  hipLaunchKernelGGL(pairs_to_table_kernel, dim3(sf_grid((long long)K * L, kThreads)),
                     dim3(kThreads), 0, st, indice_pairs, indice_num, K, L, nbr, nbr_stride, n_out);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

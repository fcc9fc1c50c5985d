// Shared device/host helpers for the sparsefusion HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/sparsefusion_hip.h"

#define SF_EMPTY_KEY 0xFFFFFFFFu
#define SF_INT_MAX 0x7FFFFFFF
#define SF_WAVE 64

void sf_set_error(const char *fmt, ...);
// name (template arguments included, as a rocprofv3 kernel trace prints it) of the convolution
// kernel this thread launched last: sf_last_conv_kernel()
void sf_note_kernel(const char *fmt, ...);

#define SF_CHECK_ARG(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      sf_set_error(__VA_ARGS__);         \
      return SF_ERR_BAD_ARG;             \
    }                                    \
  } while (0)

#define SF_CHECK_HIP(expr)                                                        \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      sf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return SF_ERR_HIP;                                                          \
    }                                                                             \
  } while (0)

#define SF_CHECK_LAUNCH() SF_CHECK_HIP(hipGetLastError())

static inline int sf_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// Grid for a grid-stride kernel over `items` with `threads` per block: enough blocks to fill
// 256 CUs x 8 blocks, never more than the work needs, never zero.
static inline int sf_grid(long long items, int threads) {
  long long b = (items + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

// Fill `n` 32-bit words with `v`, as a KERNEL.  The library issues no hipMemset*: a memset node
// between kernel nodes of a captured linear HIP graph was not ordered with them on ROCm 7.2 (replays
// of sf_dense_bev kept marks of the previous frame), so every sequence of this
// library captures as kernel nodes only.
static __global__ void sf_fill_u32_kernel(uint32_t *__restrict__ p, size_t n, uint32_t v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    p[i] = v;
}
#define SF_FILL_U32(ptr, n_words, value, st)                                                      \
  do {                                                                                            \
    if ((size_t)(n_words) > 0) {                                                                  \
      hipLaunchKernelGGL(sf_fill_u32_kernel, dim3(sf_grid((long long)(n_words), 256)), dim3(256), \
                         0, st, (uint32_t *)(ptr), (size_t)(n_words), (uint32_t)(value));         \
      SF_CHECK_LAUNCH();                                                                          \
    }                                                                                             \
  } while (0)

// Workgroups of `kern` that are resident at once on the CURRENT device (CUs x occupancy), the grid
// of the persistent kernels.  Looked up once per device ordinal: a process may drive several GPUs,
// and both the occupancy and the opt-in to more than 48 KB of dynamic LDS belong to a device.
constexpr int kSfMaxDevices = 64;
struct SfSlotCache {
  int slots[kSfMaxDevices];
};
template <typename Kern>
static inline int sf_resident_slots(SfSlotCache &cache, Kern kern, int threads, size_t dyn_lds,
                                    size_t lds_per_group = 0) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kSfMaxDevices) return -1;
  if (cache.slots[dev] > 0) return cache.slots[dev];
  if (dyn_lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds) != hipSuccess)
    return -1;
  int cus = 256, per_cu = 1;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
    cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, dyn_lds) != hipSuccess ||
      per_cu < 1)
    per_cu = 1;
  if (lds_per_group > 0) {   // static LDS the occupancy query may not see
    int by_lds = (int)((160 * 1024) / lds_per_group);
    if (per_cu > by_lds) per_cu = by_lds < 1 ? 1 : by_lds;
  }
  cache.slots[dev] = cus * per_cu;
  return cache.slots[dev];
}

// ---------------------------------------------------------------- device side
__device__ __forceinline__ uint32_t sf_hash32(uint32_t k) {
  // multiplicative (Fibonacci) hash followed by a xor-fold; capacity is a power of two
  k *= 0x9E3779B1u;
  return k ^ (k >> 15);
}

// Scattering hash (the voxelizer's own slot table): first slot, then linear probing by one.
__device__ __forceinline__ uint32_t sf_scatter_slot(uint32_t key, uint32_t mask) {
  return sf_hash32(key) & mask;
}

// Coordinate hash of the rulebooks.  Keys are linear cell indices ((b*D + z)*H + y)*W + x; the four
// keys that share key >> 2 (four x neighbours of one (b, z, y) row whenever W % 4 == 0) share a
// GROUP of four consecutive slots (16 bytes of the key array), so that the x - 1 / x / x + 1
// neighbours of a cell are found with one or two 16-byte loads instead of three probes
// (subm27_table_kernel).  slot = 4 * ((h + probe) & (groups - 1)) + ((key + (h >> 30)) & 3) with
// h = hash(key >> 2): probing moves by whole groups and keeps the position inside the group, and
// the position is rotated by two hash bits so that inputs whose keys are all congruent mod 4 still
// use every slot.  (Round 1 measured a locality-preserving hash with probing by ONE slot: the
// neighbours' slots then form the probe sequence, primary clustering, inserts 2.7x slower.)
// capacity = mask + 1 is a power of two >= 4.
__device__ __forceinline__ uint32_t sf_group_slot(uint32_t h, uint32_t key, uint32_t mask) {
  return ((h & (mask >> 2)) << 2) | ((key + (h >> 30)) & 3u);
}
__device__ __forceinline__ uint32_t sf_first_slot(uint32_t key, uint32_t mask) {
  return sf_group_slot(sf_hash32(key >> 2), key, mask);
}
__device__ __forceinline__ uint32_t sf_next_slot(uint32_t slot, uint32_t mask) {
  return (slot + 4u) & mask;
}

// Insert `key` into an open-addressing table of `mask+1` slots; returns the slot that holds
// the key, or -1 if every slot the key may use is taken (caller raises the overflow flag).
__device__ __forceinline__ int sf_hash_insert(uint32_t *keys, uint32_t mask, uint32_t key) {
  uint32_t slot = sf_first_slot(key, mask);
  for (uint32_t probe = 0; probe <= (mask >> 2); ++probe) {
    uint32_t cur = keys[slot];
    if (cur == key) return (int)slot;
    if (cur == SF_EMPTY_KEY) {
      uint32_t prev = atomicCAS(&keys[slot], SF_EMPTY_KEY, key);
      if (prev == SF_EMPTY_KEY || prev == key) return (int)slot;
    }
    slot = sf_next_slot(slot, mask);
  }
  return -1;
}

// Look `key` up; returns the slot or -1.  The table is read-only while lookups run.
__device__ __forceinline__ int sf_hash_find(const uint32_t *__restrict__ keys, uint32_t mask,
                                            uint32_t key) {
  uint32_t slot = sf_first_slot(key, mask);
  for (uint32_t probe = 0; probe <= (mask >> 2); ++probe) {
    uint32_t cur = keys[slot];
    if (cur == key) return (int)slot;
    if (cur == SF_EMPTY_KEY) return -1;
    slot = sf_next_slot(slot, mask);
  }
  return -1;
}

// Look `key` up and return its value or -1.  Key and value of the probed slot are loaded
// together (two independent loads in flight) so a hit costs one memory latency, not two.
__device__ __forceinline__ int sf_hash_lookup(const uint32_t *__restrict__ keys,
                                              const int *__restrict__ vals, uint32_t mask,
                                              uint32_t key) {
  uint32_t slot = sf_first_slot(key, mask);
  for (uint32_t probe = 0; probe <= (mask >> 2); ++probe) {
    uint32_t cur = keys[slot];
    int v = vals[slot];
    if (cur == key) return v;
    if (cur == SF_EMPTY_KEY) return -1;
    slot = sf_next_slot(slot, mask);
  }
  return -1;
}

__device__ __forceinline__ int sf_wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// Exclusive scan of one int per thread over a block of up to 1024 threads (multiple of 64).
// `smem` needs 17 ints.  Returns the exclusive prefix; *total receives the block sum.
__device__ __forceinline__ int sf_block_excl_scan(int v, int *smem, int *total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  int incl = sf_wave_incl_scan(v);
  __syncthreads();  // smem may still be read from a previous call
  if (lane == 63) smem[wid] = incl;
  __syncthreads();
  if (wid == 0) {
    int w = (lane < nw) ? smem[lane] : 0;
    int wi = sf_wave_incl_scan(w);
    if (lane < nw) smem[lane] = wi - w;
    if (lane == nw - 1) smem[16] = wi;
  }
  __syncthreads();
  *total = smem[16];
  return incl - v + smem[wid];
}

__device__ __forceinline__ int sf_block_sum(int v, int *smem) {
  int t;
  (void)sf_block_excl_scan(v, smem, &t);
  return t;
}

// Sum of bsum[0..b) and of bsum[0..nb) computed cooperatively by the block (nb is small:
// a few hundred).  smem needs 17 ints.
__device__ __forceinline__ void sf_block_prefix_of(const int *__restrict__ bsum, int nb, int b,
                                                   int *smem, int *before, int *all) {
  int pb = 0, pa = 0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) {
    int v = bsum[i];
    pa += v;
    if (i < b) pb += v;
  }
  *before = sf_block_sum(pb, smem);
  *all = sf_block_sum(pa, smem);
}

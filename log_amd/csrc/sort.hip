// sort.hip -- A4: per-tile depth sort.  Replaces the third-party package's global 64-bit radix sort of
// (tile|depth) keys + identifyTileRanges: tiles are already separated by the bucket fill, so each tile's
// (depth,id) keys are sorted independently -- one workgroup per tile, keys staged in LDS, bitonic
// network with ascending-only comparators (so no padding is needed for non-power-of-two lengths).
// Keys are unique (the Gaussian index is the low word), hence the order is total and equals the
// stable (tile, depth) order of index-ordered input.  Output: point_list[I] = Gaussian ids.
#include "common.hpp"

template <typename Arr>
LR_DEV void lr_cmpswap(Arr s, uint32_t i, uint32_t l) {
  uint64_t a = s[i], b = s[l];
  if (a > b) { s[i] = b; s[l] = a; }
}

// Sorts s[0..L) ascending; all 256 threads of the workgroup participate.
template <typename Arr>
LR_DEV void lr_bitonic(Arr s, uint32_t L, uint32_t tid) {
  uint32_t P2 = 1;
  while (P2 < L) P2 <<= 1;
  const uint32_t pairs = P2 >> 1;
  for (uint32_t k = 2; k <= P2; k <<= 1) {
    const uint32_t half = k >> 1;
    // flip stage: i in the lower half of each k-block against its mirror image
    for (uint32_t t = tid; t < pairs; t += 256) {
      uint32_t off = t & (half - 1);
      uint32_t blk = (t - off) << 1;  // (t / half) * k
      uint32_t i = blk + off, l = blk + (k - 1 - off);
      if (l < L) lr_cmpswap(s, i, l);
    }
    __syncthreads();
    for (uint32_t j = half >> 1; j > 0; j >>= 1) {
      for (uint32_t t = tid; t < pairs; t += 256) {
        uint32_t lowbits = t & (j - 1);
        uint32_t i = ((t - lowbits) << 1) | lowbits, l = i + j;
        if (l < L) lr_cmpswap(s, i, l);
      }
      __syncthreads();
    }
  }
}

// Tiles with lo < L <= CAP: LDS path.
template <int CAP>
__global__ void __launch_bounds__(256)
lr_sort_lds_kernel(const uint32_t* __restrict__ state, uint32_t tiles, const uint64_t* __restrict__ keys,
                   uint32_t* __restrict__ plist, uint32_t lo, uint32_t capacity) {
  __shared__ __attribute__((aligned(16))) uint64_t s[CAP];
  if (state[LR_HDR_NUM] > capacity) return;
  const uint32_t* offsets = state + lr_offsets_off(tiles);
  uint32_t tile = blockIdx.x;
  uint32_t beg = offsets[tile], L = offsets[tile + 1] - beg;
  if (L <= lo || L > (uint32_t)CAP) return;
  uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < L; i += 256) s[i] = keys[beg + i];
  __syncthreads();
  lr_bitonic(s, L, tid);
  for (uint32_t i = tid; i < L; i += 256) plist[beg + i] = (uint32_t)s[i];
}

// Tiles with L > lo: same network directly on the tile's slice of the key buffer (global memory; the
// workgroup barrier orders the passes -- all traffic stays inside one CU's L1/L2 path).  Rare: only
// tiles holding more than 8192 Gaussians.
__global__ void __launch_bounds__(256)
lr_sort_global_kernel(const uint32_t* __restrict__ state, uint32_t tiles, uint64_t* keys,
                      uint32_t* __restrict__ plist, uint32_t lo, uint32_t capacity) {
  if (state[LR_HDR_NUM] > capacity) return;
  const uint32_t* offsets = state + lr_offsets_off(tiles);
  uint32_t tile = blockIdx.x;
  uint32_t beg = offsets[tile], L = offsets[tile + 1] - beg;
  if (L <= lo) return;
  uint32_t tid = threadIdx.x;
  volatile uint64_t* s = keys + beg;
  lr_bitonic(s, L, tid);
  for (uint32_t i = tid; i < L; i += 256) plist[beg + i] = (uint32_t)s[i];
}

// Size classes: the LDS footprint (8 B/key) sets how many workgroups a CU can hold, and the network is
// barrier-latency bound, so small lists must not pay for the largest class's 64 KB.
#define LR_SORT_CAP0 512    //  4 KB
#define LR_SORT_CAP1 2048   // 16 KB
#define LR_SORT_CAP2 8192   // 64 KB

void lr_launch_sort(const uint32_t* state, uint32_t tiles, uint64_t* keys, uint32_t* plist, uint32_t capacity,
                    hipStream_t s) {
  if (tiles == 0) return;
  lr_prof_begin(LRK_SORT_SMALL, s);
  hipLaunchKernelGGL(lr_sort_lds_kernel<LR_SORT_CAP0>, dim3(tiles), dim3(256), 0, s, state, tiles, keys, plist,
                     0u, capacity);
  lr_prof_end(LRK_SORT_SMALL, s);
  lr_prof_begin(LRK_SORT_LARGE, s);
  hipLaunchKernelGGL(lr_sort_lds_kernel<LR_SORT_CAP1>, dim3(tiles), dim3(256), 0, s, state, tiles, keys, plist,
                     (uint32_t)LR_SORT_CAP0, capacity);
  hipLaunchKernelGGL(lr_sort_lds_kernel<LR_SORT_CAP2>, dim3(tiles), dim3(256), 0, s, state, tiles, keys, plist,
                     (uint32_t)LR_SORT_CAP1, capacity);
  lr_prof_end(LRK_SORT_LARGE, s);
  lr_prof_begin(LRK_SORT_HUGE, s);
  hipLaunchKernelGGL(lr_sort_global_kernel, dim3(tiles), dim3(256), 0, s, state, tiles, keys, plist,
                     (uint32_t)LR_SORT_CAP2, capacity);
  lr_prof_end(LRK_SORT_HUGE, s);
}

// sort.hip -- A4: per-tile depth sort.  Replaces the third-party package's global 64-bit radix sort of
// (tile|depth) keys + identifyTileRanges: tiles are already separated by the bucket fill, so each tile's
// (depth,id) keys are sorted independently -- one workgroup per tile, keys staged in LDS, register-blocked
// bitonic network with ascending-only comparators (lists are padded with +inf to a power of two).
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

// ---- register-blocked LDS path -----------------------------------------------------------------------------
// Tiles with lo < L <= hi.  Keys live in LDS (padded to a power of two with +inf), but every thread pulls EIGHT
// keys into registers per visit and runs up to THREE network levels on them before they go back, so a list
// of 2048 keys needs 23 LDS round trips / barriers instead of 66 (one per level).
//   * levels 2,4,8: sort the 8 consecutive keys of an item in registers;
//   * phase k >= 16: the "flip" level (i <-> mirror of i inside its k-block) is fused with the k/4 and k/8
//     half-cleaners -- the 8 keys {x ^ s : s in span(k-1, k/4, k/8)} are closed under all three;
//   * remaining half-cleaners in groups of three strides (8 keys at base + m*stride), the last group being the
//     thread's 8 consecutive keys.
// Registers are always ordered by element index, so every compare-exchange is "min to the lower register".
// LDS index i is stored at i + (i >> 3): one pad slot per 8 keys makes both the blocked accesses (lane stride 8
// keys -> 9) and the strided ones conflict-free for ds_read_b64/ds_write_b64.
LR_DEV uint32_t lr_phys(uint32_t i) { return i + (i >> 3); }
LR_DEV void lr_cx(uint64_t& a, uint64_t& b) {
  const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
  a = lo; b = hi;
}
LR_DEV void lr_levels_421(uint64_t r[8]) {  // strides 4, 2, 1 over the 8 registers
  lr_cx(r[0], r[4]); lr_cx(r[1], r[5]); lr_cx(r[2], r[6]); lr_cx(r[3], r[7]);
  lr_cx(r[0], r[2]); lr_cx(r[1], r[3]); lr_cx(r[4], r[6]); lr_cx(r[5], r[7]);
  lr_cx(r[0], r[1]); lr_cx(r[2], r[3]); lr_cx(r[4], r[5]); lr_cx(r[6], r[7]);
}
LR_DEV void lr_flip_421(uint64_t r[8]) {  // mirror level, then strides 2, 1
  lr_cx(r[0], r[7]); lr_cx(r[1], r[6]); lr_cx(r[2], r[5]); lr_cx(r[3], r[4]);
  lr_cx(r[0], r[2]); lr_cx(r[1], r[3]); lr_cx(r[4], r[6]); lr_cx(r[5], r[7]);
  lr_cx(r[0], r[1]); lr_cx(r[2], r[3]); lr_cx(r[4], r[5]); lr_cx(r[6], r[7]);
}
LR_DEV void lr_sort8(uint64_t r[8]) {  // phases k = 2, 4, 8 of the same network
  lr_cx(r[0], r[1]); lr_cx(r[2], r[3]); lr_cx(r[4], r[5]); lr_cx(r[6], r[7]);
  lr_cx(r[0], r[3]); lr_cx(r[1], r[2]); lr_cx(r[4], r[7]); lr_cx(r[5], r[6]);
  lr_cx(r[0], r[1]); lr_cx(r[2], r[3]); lr_cx(r[4], r[5]); lr_cx(r[6], r[7]);
  lr_flip_421(r);
}

template <int NT>
__global__ void __launch_bounds__(NT)
lr_sort_rb_kernel(const uint32_t* __restrict__ state, uint32_t tiles, const uint64_t* __restrict__ keys,
                  uint32_t* __restrict__ plist, uint32_t lo, uint32_t hi, uint32_t capacity) {
  extern __shared__ __attribute__((aligned(16))) uint64_t s[];
  if (state[LR_HDR_NUM] > capacity) return;
  const uint32_t* offsets = state + lr_offsets_off(tiles);
  const uint32_t tile = blockIdx.x;
  const uint32_t beg = offsets[tile], L = offsets[tile + 1] - beg;
  if (L <= lo || L > hi) return;
  const uint32_t tid = threadIdx.x;
  uint32_t P2 = 8;
  while (P2 < L) P2 <<= 1;
  const uint32_t nitems = P2 >> 3;
  for (uint32_t i = tid; i < P2; i += NT) s[lr_phys(i)] = i < L ? keys[beg + i] : ~0ull;
  __syncthreads();
  uint64_t r[8];
  for (uint32_t it = tid; it < nitems; it += NT) {
#pragma unroll
    for (int m = 0; m < 8; m++) r[m] = s[lr_phys(8 * it + m)];
    lr_sort8(r);
#pragma unroll
    for (int m = 0; m < 8; m++) s[lr_phys(8 * it + m)] = r[m];
  }
  __syncthreads();
  for (uint32_t k = 16, p = 3; k <= P2; k <<= 1, p++) {  // k/2 == 1 << p
    {  // flip level fused with the k/4 and k/8 half-cleaners
      const uint32_t q = p - 2, st = k >> 3;
      for (uint32_t it = tid; it < nitems; it += NT) {
        const uint32_t x = ((it >> q) << (p + 1)) | (it & ((1u << q) - 1u));
        const uint32_t y = x ^ (k - 1u) ^ (k >> 2) ^ (k >> 3);
#pragma unroll
        for (int m = 0; m < 4; m++) { r[m] = s[lr_phys(x + m * st)]; r[4 + m] = s[lr_phys(y + m * st)]; }
        lr_flip_421(r);
#pragma unroll
        for (int m = 0; m < 4; m++) { s[lr_phys(x + m * st)] = r[m]; s[lr_phys(y + m * st)] = r[4 + m]; }
      }
      __syncthreads();
    }
    for (int e = (int)p - 3; e >= 0;) {  // remaining half-cleaners: strides 2^e ... 1
      if (e >= 2) {
        const uint32_t q = (uint32_t)e - 2u;
        for (uint32_t it = tid; it < nitems; it += NT) {
          const uint32_t base = ((it >> q) << (q + 3)) | (it & ((1u << q) - 1u));
#pragma unroll
          for (int m = 0; m < 8; m++) r[m] = s[lr_phys(base + ((uint32_t)m << q))];
          lr_levels_421(r);
#pragma unroll
          for (int m = 0; m < 8; m++) s[lr_phys(base + ((uint32_t)m << q))] = r[m];
        }
        e -= 3;
      } else {
        for (uint32_t it = tid; it < nitems; it += NT) {
#pragma unroll
          for (int m = 0; m < 8; m++) r[m] = s[lr_phys(8 * it + m)];
          if (e == 1) { lr_cx(r[0], r[2]); lr_cx(r[1], r[3]); lr_cx(r[4], r[6]); lr_cx(r[5], r[7]); }
          lr_cx(r[0], r[1]); lr_cx(r[2], r[3]); lr_cx(r[4], r[5]); lr_cx(r[6], r[7]);
#pragma unroll
          for (int m = 0; m < 8; m++) s[lr_phys(8 * it + m)] = r[m];
        }
        e = -1;
      }
      __syncthreads();
    }
  }
  for (uint32_t i = tid; i < L; i += NT) plist[beg + i] = (uint32_t)s[lr_phys(i)];
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

// Size classes: the LDS footprint (9 B/key with padding) sets how many workgroups a CU can hold, so small
// lists must not pay for the largest class.
#define LR_SORT_CAP0 512    // 64 threads (one wave), 4.5 KB
#define LR_SORT_CAP1 2048   // 256 threads, 18 KB
#define LR_SORT_CAP2 8192   // 256 threads, 72 KB (dynamic LDS beyond the 64 KB static limit)
static inline size_t lr_sort_lds_bytes(uint32_t cap) { return sizeof(uint64_t) * (size_t)(cap + (cap >> 3)); }

void lr_launch_sort(const uint32_t* state, uint32_t tiles, uint64_t* keys, uint32_t* plist, uint32_t capacity,
                    hipStream_t s) {
  if (tiles == 0) return;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lr_sort_rb_kernel<256>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lr_sort_lds_bytes(LR_SORT_CAP2));
    attr_set = true;
  }
  lr_prof_begin(LRK_SORT_SMALL, s);
  hipLaunchKernelGGL(lr_sort_rb_kernel<64>, dim3(tiles), dim3(64), lr_sort_lds_bytes(LR_SORT_CAP0), s, state, tiles,
                     keys, plist, 0u, (uint32_t)LR_SORT_CAP0, capacity);
  lr_prof_end(LRK_SORT_SMALL, s);
  lr_prof_begin(LRK_SORT_LARGE, s);
  hipLaunchKernelGGL(lr_sort_rb_kernel<256>, dim3(tiles), dim3(256), lr_sort_lds_bytes(LR_SORT_CAP1), s, state, tiles,
                     keys, plist, (uint32_t)LR_SORT_CAP0, (uint32_t)LR_SORT_CAP1, capacity);
  hipLaunchKernelGGL(lr_sort_rb_kernel<256>, dim3(tiles), dim3(256), lr_sort_lds_bytes(LR_SORT_CAP2), s, state, tiles,
                     keys, plist, (uint32_t)LR_SORT_CAP1, (uint32_t)LR_SORT_CAP2, capacity);
  lr_prof_end(LRK_SORT_LARGE, s);
  lr_prof_begin(LRK_SORT_HUGE, s);
  hipLaunchKernelGGL(lr_sort_global_kernel, dim3(tiles), dim3(256), 0, s, state, tiles, keys, plist,
                     (uint32_t)LR_SORT_CAP2, capacity);
  lr_prof_end(LRK_SORT_HUGE, s);
}

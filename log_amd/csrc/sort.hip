// sort.hip -- A4: per-tile depth sort.  Replaces the third-party package's global 64-bit radix sort of
// (tile|depth) keys + identifyTileRanges: tiles are already separated by the bucket fill, so each tile's
// (depth,id) keys are sorted independently -- one workgroup per tile, keys staged in LDS, register-blocked
// bitonic network with ascending-only comparators (lists are padded with +inf to a power of two).
// Keys are unique (the Gaussian index is the low word), hence the order is total and equals the
// stable (tile, depth) order of index-ordered input.  Output: point_list[I] = Gaussian ids.  Lists longer than one
// LDS block (8192 keys) take the hybrid multi-block path at the end of this file.
#include "common.hpp"

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

// ---- LDS building blocks (s[] holds P2 keys at padded positions; all NT threads of the workgroup call) ---------
// Half-cleaner levels with strides 2^e_top ... 1, three per LDS round trip.
template <int NT>
LR_DEV void lr_lds_halfcleaners(uint64_t* s, uint32_t nitems, uint32_t tid, int e_top) {
  uint64_t r[8];
  for (int e = e_top; e >= 0;) {
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

// Full ascending sort of the P2 (power of two, >= 8) keys in s[].
template <int NT>
LR_DEV void lr_lds_sort(uint64_t* s, uint32_t P2, uint32_t tid) {
  const uint32_t nitems = P2 >> 3;
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
    lr_lds_halfcleaners<NT>(s, nitems, tid, (int)p - 3);  // remaining strides k/16 ... 1
  }
}

// Tiles with lo < L <= hi: the whole list in one workgroup's LDS.
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
  for (uint32_t i = tid; i < P2; i += NT) s[lr_phys(i)] = i < L ? keys[beg + i] : ~0ull;
  __syncthreads();
  lr_lds_sort<NT>(s, P2, tid);
  for (uint32_t i = tid; i < L; i += NT) plist[beg + i] = (uint32_t)s[lr_phys(i)];
}

// ---- lists longer than LR_SORT_BLOCK: hybrid network ------------------------------------------------------------
// The same ascending-only bitonic network over the tile's whole list, split by stride: levels whose stride is
// >= LR_SORT_BLOCK are single streaming passes over the tile's slice of the key buffer in global memory (one
// compare-exchange per pair, every workgroup of the launch works on its share of pairs); everything below that
// stride stays inside an LR_SORT_BLOCK-key block and runs in LDS with the register-blocked code above.  A list of
// B*2^m keys costs m(m+1)/2 global passes + m+1 LDS passes instead of ~log^2 passes through global memory
// (measured before: 27 ms per view at 30 M Gaussians).  Launches are ordered by the stream; tiles that do not take
// part in a level (list too short) exit at once.  blockIdx.x indexes the big-tile list written by the scan kernel.
struct LrBigTile { uint32_t beg, L, P2; bool ok; };
LR_DEV LrBigTile lr_big_tile(const uint32_t* __restrict__ state, uint32_t tiles, uint32_t capacity) {
  LrBigTile t{0u, 0u, 0u, false};
  if (state[LR_HDR_NUM] > capacity || blockIdx.x >= state[LR_HDR_NBIG]) return t;
  const uint32_t tile = state[lr_biglist_off(tiles) + blockIdx.x];
  const uint32_t* offsets = state + lr_offsets_off(tiles);
  t.beg = offsets[tile];
  t.L = offsets[tile + 1] - t.beg;
  t.P2 = LR_SORT_BLOCK;
  while (t.P2 < t.L) t.P2 <<= 1;
  t.ok = true;
  return t;
}

// Stage 0: every LR_SORT_BLOCK-key block of a big tile sorted on its own (blockIdx.y = block).
template <int NT>
__global__ void __launch_bounds__(NT)
lr_bigsort_blocks_kernel(const uint32_t* __restrict__ state, uint32_t tiles, uint64_t* __restrict__ keys,
                         uint32_t capacity) {
  extern __shared__ __attribute__((aligned(16))) uint64_t s[];
  const LrBigTile t = lr_big_tile(state, tiles, capacity);
  if (!t.ok) return;
  const uint32_t b0 = blockIdx.y * LR_SORT_BLOCK;
  if (b0 >= t.L) return;
  const uint32_t cnt = min((uint32_t)LR_SORT_BLOCK, t.L - b0), tid = threadIdx.x;
  uint32_t P2 = 8;
  while (P2 < cnt) P2 <<= 1;
  uint64_t* k = keys + t.beg + b0;
  for (uint32_t i = tid; i < P2; i += NT) s[lr_phys(i)] = i < cnt ? k[i] : ~0ull;
  __syncthreads();
  lr_lds_sort<NT>(s, P2, tid);
  for (uint32_t i = tid; i < cnt; i += NT) k[i] = s[lr_phys(i)];
}

// One global level of phase k: the flip level (j == 0) or the half-cleaner with stride j >= LR_SORT_BLOCK.
__global__ void __launch_bounds__(256)
lr_bigsort_global_kernel(const uint32_t* __restrict__ state, uint32_t tiles, uint64_t* __restrict__ keys,
                         uint32_t capacity, uint32_t k, uint32_t j) {
  const LrBigTile t = lr_big_tile(state, tiles, capacity);
  if (!t.ok || (k >> 1) >= t.L) return;  // phase k only merges something when the list reaches past k/2
  uint64_t* a = keys + t.beg;
  const uint32_t pairs = t.P2 >> 1;
  for (uint32_t p = blockIdx.y * 256 + threadIdx.x; p < pairs; p += gridDim.y * 256) {
    uint32_t i, l;
    if (j == 0) {
      const uint32_t half = k >> 1, off = p & (half - 1u), blk = (p - off) << 1;
      i = blk + off; l = blk + (k - 1u - off);
    } else {
      const uint32_t low = p & (j - 1u);
      i = ((p - low) << 1) | low; l = i + j;
    }
    if (l < t.L) {
      const uint64_t x = a[i], y = a[l];
      if (x > y) { a[i] = y; a[l] = x; }
    }
  }
}

// Tail of phase k: all strides below LR_SORT_BLOCK, block by block in LDS (blockIdx.y = block).
template <int NT>
__global__ void __launch_bounds__(NT)
lr_bigsort_tail_kernel(const uint32_t* __restrict__ state, uint32_t tiles, uint64_t* __restrict__ keys,
                       uint32_t capacity, uint32_t k) {
  extern __shared__ __attribute__((aligned(16))) uint64_t s[];
  const LrBigTile t = lr_big_tile(state, tiles, capacity);
  if (!t.ok || (k >> 1) >= t.L) return;
  const uint32_t b0 = blockIdx.y * LR_SORT_BLOCK;
  if (b0 >= t.L) return;
  const uint32_t cnt = min((uint32_t)LR_SORT_BLOCK, t.L - b0), tid = threadIdx.x;
  uint64_t* kk = keys + t.beg + b0;
  for (uint32_t i = tid; i < LR_SORT_BLOCK; i += NT) s[lr_phys(i)] = i < cnt ? kk[i] : ~0ull;
  __syncthreads();
  lr_lds_halfcleaners<NT>(s, LR_SORT_BLOCK >> 3, tid, 12);  // strides 4096 ... 1  (LR_SORT_BLOCK == 8192)
  for (uint32_t i = tid; i < cnt; i += NT) kk[i] = s[lr_phys(i)];
}

__global__ void __launch_bounds__(256)
lr_bigsort_emit_kernel(const uint32_t* __restrict__ state, uint32_t tiles, const uint64_t* __restrict__ keys,
                       uint32_t* __restrict__ plist, uint32_t capacity) {
  const LrBigTile t = lr_big_tile(state, tiles, capacity);
  if (!t.ok) return;
  for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < t.L; i += gridDim.y * 256)
    plist[t.beg + i] = (uint32_t)keys[t.beg + i];
}

// Size classes: the LDS footprint (9 B/key with padding) sets how many workgroups a CU can hold, so small
// lists must not pay for the largest class.
#define LR_SORT_CAP0 512             // 64 threads (one wave), 4.5 KB
#define LR_SORT_CAP1 2048            // 256 threads, 18 KB
#define LR_SORT_CAP2 LR_SORT_BLOCK   // 256 threads, 72 KB (dynamic LDS beyond the 64 KB static limit)
static_assert(LR_SORT_BLOCK == 8192, "lr_bigsort_tail_kernel hard-codes the top stride exponent");
static inline size_t lr_sort_lds_bytes(uint32_t cap) { return sizeof(uint64_t) * (size_t)(cap + (cap >> 3)); }

// max_len: upper bound on the longest tile list known to the HOST (exact count from stage 1, a hint in sync-free
// operation, or 0 = unknown -> assume `capacity`).  It only decides how many multi-block levels are launched.
void lr_launch_sort(const uint32_t* state, uint32_t tiles, uint64_t* keys, uint32_t* plist, uint32_t capacity,
                    uint32_t max_len, hipStream_t s) {
  if (tiles == 0) return;
  static bool attr_set = false;
  if (!attr_set) {
    const int big = (int)lr_sort_lds_bytes(LR_SORT_BLOCK);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lr_sort_rb_kernel<256>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lr_bigsort_blocks_kernel<512>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lr_bigsort_tail_kernel<512>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, big);
    attr_set = true;
  }
  if (max_len == 0 || max_len > capacity) max_len = capacity;
  lr_prof_begin(LRK_SORT_SMALL, s);
  hipLaunchKernelGGL(lr_sort_rb_kernel<64>, dim3(tiles), dim3(64), lr_sort_lds_bytes(LR_SORT_CAP0), s, state, tiles,
                     keys, plist, 0u, (uint32_t)LR_SORT_CAP0, capacity);
  lr_prof_end(LRK_SORT_SMALL, s);
  if (max_len > LR_SORT_CAP0) {
    lr_prof_begin(LRK_SORT_LARGE, s);
    hipLaunchKernelGGL(lr_sort_rb_kernel<256>, dim3(tiles), dim3(256), lr_sort_lds_bytes(LR_SORT_CAP1), s, state, tiles,
                       keys, plist, (uint32_t)LR_SORT_CAP0, (uint32_t)LR_SORT_CAP1, capacity);
    if (max_len > LR_SORT_CAP1)
      hipLaunchKernelGGL(lr_sort_rb_kernel<256>, dim3(tiles), dim3(256), lr_sort_lds_bytes(LR_SORT_CAP2), s, state,
                         tiles, keys, plist, (uint32_t)LR_SORT_CAP1, (uint32_t)LR_SORT_CAP2, capacity);
    lr_prof_end(LRK_SORT_LARGE, s);
  }
  if (max_len > LR_SORT_BLOCK) {
    // every big tile holds more than LR_SORT_BLOCK keys, so there are at most capacity / LR_SORT_BLOCK of them
    const uint32_t nbig = min(tiles, capacity / LR_SORT_BLOCK + 1u);
    const uint32_t nblk = (max_len + LR_SORT_BLOCK - 1u) / LR_SORT_BLOCK;
    const uint32_t ypass = min(64u, max(1u, nblk * 4u));  // workgroups per tile for the streaming passes
    const size_t lds = lr_sort_lds_bytes(LR_SORT_BLOCK);
    lr_prof_begin(LRK_SORT_HUGE, s);
    hipLaunchKernelGGL(lr_bigsort_blocks_kernel<512>, dim3(nbig, nblk), dim3(512), lds, s, state, tiles, keys, capacity);
    for (uint64_t k = 2ull * LR_SORT_BLOCK; (k >> 1) < max_len; k <<= 1) {
      hipLaunchKernelGGL(lr_bigsort_global_kernel, dim3(nbig, ypass), dim3(256), 0, s, state, tiles, keys, capacity,
                         (uint32_t)k, 0u);
      for (uint64_t j = k >> 2; j >= LR_SORT_BLOCK; j >>= 1)
        hipLaunchKernelGGL(lr_bigsort_global_kernel, dim3(nbig, ypass), dim3(256), 0, s, state, tiles, keys, capacity,
                           (uint32_t)k, (uint32_t)j);
      hipLaunchKernelGGL(lr_bigsort_tail_kernel<512>, dim3(nbig, nblk), dim3(512), lds, s, state, tiles, keys, capacity,
                         (uint32_t)k);
    }
    hipLaunchKernelGGL(lr_bigsort_emit_kernel, dim3(nbig, ypass), dim3(256), 0, s, state, tiles, keys, plist, capacity);
    lr_prof_end(LRK_SORT_HUGE, s);
  }
}

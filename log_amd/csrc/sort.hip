// sort.hip -- A4: per-tile depth sort.  Replaces the third-party package's global 64-bit radix sort of
// (tile|depth) keys + identifyTileRanges: tiles are already separated by the bucket fill, so each tile's
// (depth,id) keys are sorted independently, one workgroup per tile: a depth-bucket distribution sort (O(L)), with a
// register-blocked bitonic network (ascending-only comparators, lists padded with +inf to a power of two) as the
// fallback for depths no bucket map can spread and as the LOGRAST_BUCKET_SORT=0 reference of the same total order.
// Keys are unique (the Gaussian index is the low word), hence the order is total and equals the
// stable (tile, depth) order of index-ordered input.  Output: point_list[I] = Gaussian ids.
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

// ---- the network on a list longer than one LDS block, by ONE workgroup -----------------------------------------
// The same ascending-only bitonic network over a[0, L) (virtually padded with +inf to a power of two), split by stride:
// levels whose stride is >= LR_SORT_BLOCK are passes over the list in global memory (one compare-exchange per pair; the
// 160 KB slice of a 20 K-key tile stays in L2), everything below that stride runs block by block in LDS with the
// register-blocked code above.  A list of B*2^m keys costs m(m+1)/2 global passes + (m+1) LDS passes over its blocks.
// This is the fallback of the long-list kernels for depths a bucket map cannot spread (a surface exactly parallel to
// the image plane: identical depths), so it only has to be correct and not absurd: it runs inside the workgroup that
// found the list unsortable by buckets -- no extra launches (a separate multi-workgroup version of the same passes cost
// every view eight launches of idle workgroups, 54 us at 30 M Gaussians, to be there for the rare tile that needs it).
// s: LDS, lr_sort_lds_bytes(LR_SORT_BLOCK) bytes.  All NT threads call; workgroup-scope visibility of the global
// stores between passes comes from the barriers (one workgroup = one CU = one vector L1).
template <int NT>
LR_DEV void lr_wg_hybrid_sort(uint64_t* __restrict__ a, uint32_t L, uint64_t* s, uint32_t tid) {
  uint32_t P2 = LR_SORT_BLOCK;
  while (P2 < L) P2 <<= 1;
  for (uint32_t b0 = 0; b0 < L; b0 += LR_SORT_BLOCK) {       // stage 0: every block sorted on its own
    const uint32_t cnt = min((uint32_t)LR_SORT_BLOCK, L - b0);
    uint32_t p2 = 8;
    while (p2 < cnt) p2 <<= 1;
    for (uint32_t i = tid; i < p2; i += NT) s[lr_phys(i)] = i < cnt ? a[b0 + i] : ~0ull;
    __syncthreads();
    lr_lds_sort<NT>(s, p2, tid);
    for (uint32_t i = tid; i < cnt; i += NT) a[b0 + i] = s[lr_phys(i)];
    __syncthreads();
  }
  const uint32_t pairs = P2 >> 1;
  for (uint64_t k = 2ull * LR_SORT_BLOCK; (k >> 1) < L; k <<= 1) {   // phase k merges sorted runs of k/2 keys
    const uint32_t kk = (uint32_t)k, half = kk >> 1;
    for (uint32_t p = tid; p < pairs; p += NT) {             // the flip level: i <-> mirror of i inside its k-block
      const uint32_t off = p & (half - 1u), blk = (p - off) << 1;
      const uint32_t i = blk + off, l = blk + (kk - 1u - off);
      if (l < L) { const uint64_t x = a[i], y = a[l]; if (x > y) { a[i] = y; a[l] = x; } }
    }
    __syncthreads();
    for (uint32_t j = kk >> 2; j >= LR_SORT_BLOCK; j >>= 1) { // half-cleaners with strides >= one block
      for (uint32_t p = tid; p < pairs; p += NT) {
        const uint32_t low = p & (j - 1u);
        const uint32_t i = ((p - low) << 1) | low, l = i + j;
        if (l < L) { const uint64_t x = a[i], y = a[l]; if (x > y) { a[i] = y; a[l] = x; } }
      }
      __syncthreads();
    }
    for (uint32_t b0 = 0; b0 < L; b0 += LR_SORT_BLOCK) {     // strides below one block: block by block in LDS
      const uint32_t cnt = min((uint32_t)LR_SORT_BLOCK, L - b0);
      for (uint32_t i = tid; i < LR_SORT_BLOCK; i += NT) s[lr_phys(i)] = i < cnt ? a[b0 + i] : ~0ull;
      __syncthreads();
      lr_lds_halfcleaners<NT>(s, LR_SORT_BLOCK >> 3, tid, 12);   // strides 4096 ... 1  (LR_SORT_BLOCK == 8192)
      for (uint32_t i = tid; i < cnt; i += NT) a[b0 + i] = s[lr_phys(i)];
      __syncthreads();
    }
  }
}
static_assert(LR_SORT_BLOCK == 8192, "lr_wg_hybrid_sort hard-codes the top stride exponent of a block");

// Tiles with lo < L <= hi: the whole list in one workgroup's LDS.
template <int NT>
__global__ void __launch_bounds__(NT)
lr_sort_rb_kernel(const uint32_t* __restrict__ state, uint32_t tiles, const uint64_t* __restrict__ keys,
                  uint32_t* __restrict__ plist, uint32_t lo, uint32_t hi, uint32_t capacity) {
  extern __shared__ __attribute__((aligned(16))) uint64_t s[];
  if (lr_bail(state, capacity)) return;
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

// ---- depth-bucket path -----------------------------------------------------------------------------------------
// The keys of a tile are (depth, id) with depths spread over the tile's depth range, so a distribution sort does in
// O(L) what the network does in O(L log^2 L): one workgroup per tile,
//   1. keys -> LDS, min / max depth (LDS atomics);
//   2. bucket = floor((depth - min) / (max - min) * nb), nb ~ L/2..L/4 buckets; the returning LDS atomic that counts the
//      bucket also ranks the key inside it (integer LDS atomics run at ~7 per clock per CU);
//   3. exclusive scan of the nb counts;
//   4. scatter into bucket order;
//   5. every key counts the keys of its own bucket that are smaller (full 64-bit compare: ties in depth fall in
//      the same bucket and are ordered by id) and writes its id to list position bucket_start + that count.
// Same total order, hence the same list as the network, bit for bit.  If the depths are so clustered that a bucket
// holds more than LR_BUCKET_MAX keys the workgroup falls back to the network on the keys it already staged.
#define LR_BUCKET_MAX 32
// ---- depth -> bucket map, equalised over a sample ------------------------------------------------------------------
// A linear map of [min depth, max depth] onto the buckets spends most of them on gaps when the depths cluster -- a
// foreground object in front of a background, a surface seen nearly head-on plus a few strays -- and the clusters then
// overflow their few buckets (LR_BUCKET_MAX keys) and send the tile to the network fallback.  So the range is cut into
// LR_CELLS cells, a SAMPLE of the list (its first keys: arrival order is unrelated to depth) is histogrammed over them,
// and every cell gets one bucket plus a share of the remaining buckets proportional to its sample count; inside a cell
// the map is linear.  Monotone in depth, so the order it produces is the same total order.  One extra LDS read per key.
#define LR_CELLS 64
struct LrDepthMap {
  float fmin, cscale;            // cell coordinate = (depth - fmin) * cscale
  uint32_t ncells;
  const uint32_t* table;         // [ncells] in LDS: first bucket | buckets << 16
};
LR_DEV uint32_t lr_depth_cell(const LrDepthMap& m, uint32_t dbits, float& frac) {
  const float rel = (__uint_as_float(dbits) - m.fmin) * m.cscale;
  const uint32_t c = rel >= 0.f ? (uint32_t)fminf(rel, (float)(m.ncells - 1u)) : 0u;   // below the range / NaN -> cell 0
  frac = fminf(rel - (float)c, 0.99999994f);              // inf / NaN (range == 0) -> the cell's last bucket, for every key
  frac = frac >= 0.f ? frac : 0.f;
  return c;
}
LR_DEV uint32_t lr_depth_bucket(const LrDepthMap& m, uint32_t dbits) {
  float frac;
  const uint32_t t = m.table[lr_depth_cell(m, dbits, frac)];
  const uint32_t n = t >> 16;
  return (t & 0xffffu) + min((uint32_t)(frac * (float)n), n - 1u);
}
// Built by the whole workgroup: cellcnt[LR_CELLS] must be zero and hold the sample histogram on entry (filled with
// lr_depth_cell on a map whose table is not used yet); `sampled` = keys histogrammed; nb = buckets to hand out
// (nb >= 2 * ncells).  Needs a barrier before and after.
LR_DEV void lr_depth_map_build(uint32_t* table, const uint32_t* cellcnt, uint32_t ncells, uint32_t sampled, uint32_t nb,
                                   bool equalize = true) {
  if (threadIdx.x < 64u) {                                  // one wave: LR_CELLS <= 64
    const uint32_t t = threadIdx.x;
    const uint32_t c = t < ncells ? cellcnt[t] : 0u;
    const uint32_t n = t >= ncells ? 0u : (equalize ? 1u + (uint32_t)(((uint64_t)c * (nb - ncells)) / max(sampled, 1u)) : nb / ncells);
    uint32_t inc = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(inc, d);
      if ((int)t >= d) inc += up;
    }
    if (t < ncells) table[t] = (inc - n) | (n << 16);
  }
}

// Final order inside buckets, one thread per BUCKET: consecutive threads read consecutive LDS addresses (every key
// once, no bank conflicts -- ranking every key against its bucket read each key ~6 times from random banks and was
// LDS-bandwidth bound), sort up to 8 keys in registers with the network's 8-key kernel (eight independent LDS loads:
// one round trip), and write the ids in order.
// Buckets of 9..LR_BUCKET_MAX keys (a few per cent at the usual mean of 2-5 keys, so nearly every wave meets one) are
// ranked by the WHOLE WAVE, one bucket at a time: lane l loads key l of the bucket (one round trip), every lane counts
// the smaller keys through readlane broadcasts (no further LDS access), and writes its id.  What this replaces -- the
// owning thread ranking the bucket key by key, n^2 DEPENDENT LDS reads at ~150 ns each under load -- cost 20-30 us per
// 12 K-key window (wall_clock64 per phase), i.e. most of the long-list sort; one thread per KEY, each counting the
// smaller keys of its own bucket (~20 dependent round trips per key), measured 37 us.
// Call with the whole wave converged; `valid` = this lane owns a bucket.  key_at(pos) returns the key staged for list
// position pos; out[ostride * pos] receives the id of list position pos.
template <typename KeyAt>
LR_DEV void lr_emit_bucket(KeyAt key_at, uint32_t st, uint32_t en, bool valid, uint32_t* out, uint32_t ostride = 1u) {
  const uint32_t n = valid ? en - st : 0u;
  if (n <= 8u) {
    uint64_t r[8];
#pragma unroll
    for (int m = 0; m < 8; m++) r[m] = (uint32_t)m < n ? key_at(st + (uint32_t)m) : ~0ull;
    lr_sort8(r);
#pragma unroll
    for (int m = 0; m < 8; m++)
      if ((uint32_t)m < n) out[(size_t)ostride * (st + m)] = (uint32_t)r[m];
  }
  uint64_t big = __ballot(n > 8u);
  const uint32_t lane = threadIdx.x & 63u;
  while (big) {   // two big buckets per round: their keys are requested together
    uint32_t bst[2], bn[2];
    uint64_t key[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const bool have = big != 0;
      const int src = have ? __builtin_ctzll(big) : 0;
      big &= big - 1;                                        // no-op once big == 0
      bst[q] = (uint32_t)lr_readlane_i((int)st, src);
      bn[q] = have ? min((uint32_t)lr_readlane_i((int)n, src), 64u) : 0u;   // <= LR_BUCKET_MAX by the callers' check
      key[q] = lane < bn[q] ? key_at(bst[q] + lane) : ~0ull;
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const uint32_t klo = (uint32_t)key[q], khi = (uint32_t)(key[q] >> 32);
      uint32_t smaller = 0;
      for (uint32_t j = 0; j < bn[q]; j++) {
        const uint64_t kj = ((uint64_t)(uint32_t)lr_readlane_i((int)khi, (int)j) << 32) | (uint32_t)lr_readlane_i((int)klo, (int)j);
        smaller += kj < key[q] ? 1u : 0u;
      }
      if (lane < bn[q]) out[(size_t)ostride * (bst[q] + smaller)] = klo;
    }
  }
}
// One tile's list (keys[beg, beg + L), L <= NT * KPT) by the workgroup's NT threads; s = the workgroup's dynamic LDS,
// lr_bucket_lds_bytes(NT * KPT) bytes.  The keys are also staged in network layout so that the workgroup can fall back
// to the network on them.
template <int NT, int KPT>
LR_DEV void lr_bucket_tile(const uint64_t* __restrict__ keys, uint32_t* __restrict__ plist, uint32_t beg, uint32_t L,
                           int equalize, uint64_t* s) {
  constexpr uint32_t CAP = NT * KPT;                       // longest list of this class (a power of two)
  // s: [A[CAP + CAP/8] (network layout)] | B[CAP] | cnt[CAP/4]
  uint64_t* const A = s;
  uint64_t* const Bk = s + (CAP + (CAP >> 3));
  uint32_t* const cnt = reinterpret_cast<uint32_t*>(Bk + CAP);
  __shared__ uint32_t sh_min, sh_max, sh_maxcnt, wave_tot[NT / 64], cellcnt[LR_CELLS], celltab[LR_CELLS];
  const uint32_t tid = threadIdx.x;
  uint32_t P2 = 8;
  while (P2 < L) P2 <<= 1;
  const uint32_t nb = max(P2 >> 2, 8u);                    // buckets (power of two, >= L/4; cnt[] holds CAP/4)
  if (tid == 0) { sh_min = 0xffffffffu; sh_max = 0u; sh_maxcnt = 0u; }
  if (tid < LR_CELLS) cellcnt[tid] = 0u;
  for (uint32_t b = tid; b < nb; b += NT) cnt[b] = 0u;
  uint64_t key[KPT];
  uint32_t dmin = 0xffffffffu, dmax = 0u;
#pragma unroll
  for (int k = 0; k < KPT; k++) {
    const uint32_t i = tid + (uint32_t)k * NT;
    key[k] = i < L ? keys[beg + i] : ~0ull;
    if (i < P2) A[lr_phys(i)] = key[k];                    // staged for the fallback
    if (i < L) { const uint32_t d = (uint32_t)(key[k] >> 32); dmin = min(dmin, d); dmax = max(dmax, d); }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, d));
    dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, d));
  }
  __syncthreads();
  if ((tid & 63u) == 0u) { atomicMin(&sh_min, dmin); atomicMax(&sh_max, dmax); }
  __syncthreads();
  // depth -> bucket map: first the plain linear one; if a bucket overflows, once more with the map equalised over ALL
  // keys of the list (they sit in registers; a sample of a few hundred keys over 64 cells starves cells of buckets).
  // (Equalised from the start loses on smooth distributions: a cell the list only partly covers -- the range's ends, a
  // silhouette -- concentrates its buckets' keys by the inverse of the covered fraction; measured at 30 M Gaussians,
  // a few tiles per view then took the network fallback.)
  LrDepthMap dmap;
  dmap.fmin = __uint_as_float(sh_min);
  dmap.ncells = min((uint32_t)LR_CELLS, nb >> 1);
  dmap.cscale = (float)dmap.ncells / (__uint_as_float(sh_max) - dmap.fmin);   // inf / nan when the range is 0: one bucket for all
  dmap.table = celltab;
  uint32_t bkt[KPT], rnk[KPT];
  uint32_t run = 0;
  for (int attempt = 0;; attempt++) {
    const bool eq = attempt == 1;
    if (eq) {
#pragma unroll
      for (int k = 0; k < KPT; k++) {
        const uint32_t i = tid + (uint32_t)k * NT;
        if (i < L) { float fr; atomicAdd(&cellcnt[lr_depth_cell(dmap, (uint32_t)(key[k] >> 32), fr)], 1u); }
      }
    }
    __syncthreads();
    lr_depth_map_build(celltab, cellcnt, dmap.ncells, L, nb, eq);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KPT; k++) {
      const uint32_t i = tid + (uint32_t)k * NT;
      bkt[k] = 0u; rnk[k] = 0u;
      if (i < L) {
        const uint32_t b = lr_depth_bucket(dmap, (uint32_t)(key[k] >> 32));
        bkt[k] = b;
        rnk[k] = atomicAdd(&cnt[b], 1u);
      }
    }
    __syncthreads();
    // exclusive scan of cnt[0..nb): thread t owns counters [t*per, (t+1)*per)
    const uint32_t per = (nb + NT - 1) / NT;
    uint32_t local = 0, lmax = 0;
    for (uint32_t q = 0; q < per; q++) {
      const uint32_t b = tid * per + q;
      const uint32_t c = b < nb ? cnt[b] : 0u;
      local += c; lmax = max(lmax, c);
    }
    uint32_t inc = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(inc, d);
      if ((int)(tid & 63u) >= d) inc += up;
    }
    if ((tid & 63u) == 63u) wave_tot[tid >> 6] = inc;
    atomicMax(&sh_maxcnt, lmax);
    __syncthreads();
    run = inc - local;
    for (uint32_t w = 0; w < (tid >> 6); w++) run += wave_tot[w];
    if (sh_maxcnt <= LR_BUCKET_MAX) break;
    if (eq || !equalize) {                                   // clustered beyond the map's reach: the network
      __syncthreads();
      lr_lds_sort<NT>(A, P2, tid);
      for (uint32_t i = tid; i < L; i += NT) plist[beg + i] = (uint32_t)A[lr_phys(i)];
      return;
    }
    __syncthreads();                                          // every thread has read sh_maxcnt and its counts
    if (tid == 0) sh_maxcnt = 0u;
    for (uint32_t b = tid; b < nb; b += NT) cnt[b] = 0u;
  }
  __syncthreads();                                          // every thread has read its counts
  const uint32_t per = (nb + NT - 1) / NT;
  for (uint32_t q = 0; q < per; q++) {
    const uint32_t b = tid * per + q;
    if (b < nb) { const uint32_t c = cnt[b]; cnt[b] = run; run += c; }   // cnt becomes the bucket start
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < KPT; k++) {
    const uint32_t i = tid + (uint32_t)k * NT;
    if (i < L) Bk[cnt[bkt[k]] + rnk[k]] = key[k];
  }
  __syncthreads();
  // every key counts the smaller keys of its own bucket (short lists: more threads than buckets, so one thread per
  // key beats one per bucket here -- measured both ways)
#pragma unroll
  for (int k = 0; k < KPT; k++) {
    const uint32_t i = tid + (uint32_t)k * NT;
    if (i < L) {
      const uint32_t b = bkt[k], st = cnt[b], en = (b + 1u < nb) ? cnt[b + 1u] : L;
      uint32_t smaller = 0;
      for (uint32_t j = st; j < en; j++) smaller += Bk[j] < key[k] ? 1u : 0u;
      plist[beg + st + smaller] = (uint32_t)key[k];
    }
  }
}
// Lists of up to 1024 keys: one 256-thread workgroup per tile (most tiles of a small scene).
__global__ void __launch_bounds__(256)
lr_sort_small_kernel(const uint32_t* __restrict__ state, uint32_t tiles, const uint64_t* __restrict__ keys,
                     uint32_t* __restrict__ plist, uint32_t capacity, int equalize) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lr_sort_lds[];
  if (lr_bail(state, capacity)) return;
  const uint32_t* offsets = state + lr_offsets_off(tiles);
  const uint32_t beg = offsets[blockIdx.x], L = offsets[blockIdx.x + 1] - beg;
  if (L == 0 || L > 1024u) return;
  lr_bucket_tile<256, 4>(keys, plist, beg, L, equalize, lr_sort_lds);
}
static inline size_t lr_bucket_lds_bytes(uint32_t cap) {
  return sizeof(uint64_t) * (size_t)(cap + (cap >> 3)) + sizeof(uint64_t) * cap + sizeof(uint32_t) * (cap >> 2);
}

// ---- lists above 1024 keys: one 1024-thread workgroup each, longest first ----------------------------------------
// blockIdx.x walks lr_scan_kernel's longest-first order.  Up to LR_LONG_LIST keys the list is sorted in LDS by the code
// above; beyond, the same depth-bucket sort runs with the keys streamed from memory and only the bucket counters in LDS:
//   pass 1  depth range of a sample (64-key pieces spread over the list);
//   pass 2  bucket of every key, counted with a non-returning LDS atomic; a 16-bit bucket id per key goes to the scratch
//           half of the key buffer (coalesced);
//   scan    bucket starts;
//   then, window by window (7680 list positions, cut at bucket boundaries): every key whose bucket falls into the window
//           (known from its id: 2 B re-read per key and window, cache-resident) is fetched and dropped into an LDS copy of
//           that window at the slot the bucket's start counter hands out (it doubles as the fill cursor); one thread per
//           bucket then orders the bucket's keys in registers, the ids go back into the window and leave in full lines.
// O(L) work instead of n log^2 n; nothing is scattered through memory (an earlier version scattered the keys into a
// bucket-ordered scratch copy: 8-byte stores all over a 160 KB region from 512 concurrent workgroups cost more HBM traffic
// than the whole network sort).  A list whose depths defeat both bucket maps is sorted by the network, by the same
// workgroup (lr_wg_hybrid_sort).
// (Measured alternatives, removed.  16-bit key INDICES in the window instead of the keys -- one window then covers a 20 K-
// key list and the staging runs once -- but the final order has to gather the keys back through the indices, 8-byte reads
// scattered over the tile's 176 KB slice while 500 other tiles do the same: 50 us per tile instead of 4 x 9.  One
// workgroup per tile holding the depth bits of all its keys in VGPRs -- keys read from memory once -- needs all 128 VGPRs,
// i.e. ONE workgroup per CU, and every phase of this sort is a short chain of LDS round trips at ~150 ns each: 60 us per
// 20 K-key tile against 2 x 57 us with two workgroups per CU overlapping.  Separate launches per size class -- (1024, 4096]
// in LDS with 256 threads, (4096, 8192] with the keys in registers, the rest here -- each with a grid that could reach
// every tile: at 30 M Gaussians, where every list is long, the launches that found nothing to do cost 100 us of idle
// workgroups per view, and the lists of the class in work never overlapped those of the next.)
#define LR_LONG_NB 4096     // bucket counters in LDS
#define LR_LONG_WIN_BYTES 61440   // LDS window: 7680 keys staged at a time (16 KB counters + this: two workgroups per CU)
#define LR_LONG_UNR 8       // independent loads in flight per thread in the streaming passes (64 VGPRs: two workgroups per CU, no spills)
// One list, by the whole workgroup (lr_sort_long_kernel below); blk = its position in the longest-first order.
LR_DEV void lr_sort_long_list(uint32_t* __restrict__ state, uint32_t tiles, uint64_t* __restrict__ keys,
                              uint32_t* __restrict__ ranks, uint32_t* __restrict__ plist, int equalize,
                              int network_only, int lazy, uint32_t blk) {
  constexpr uint32_t LR_LONG_WIN = LR_LONG_WIN_BYTES / sizeof(uint64_t);
  extern __shared__ uint32_t lcnt[];  // LR_LONG_NB bucket counters (then their starts) | LR_LONG_WIN + LR_BUCKET_MAX staged keys
  uint64_t* const win = reinterpret_cast<uint64_t*>(lcnt + LR_LONG_NB);
  __shared__ uint32_t sh_min, sh_max, sh_maxcnt, wave_tot[16], cellcnt[LR_CELLS], celltab[LR_CELLS];
  // blk walks the longest-first dispatch order: every tile in front of a list of more than 1024 keys holds at
  // least 1024 itself (lr_scan_kernel's length buckets), so capacity / 1024 + 1 workgroups reach all of them
  const uint32_t tile = state[lr_order_off(tiles) + blk];
  const uint32_t* offsets = state + lr_offsets_off(tiles);
  const uint32_t beg = offsets[tile], L = offsets[tile + 1] - beg, tid = threadIdx.x;
  if (L <= 1024u) return;                                   // lr_sort_small_kernel's
  // lazy (streamed lists only; common.hpp: sorted[] / open[]): 0 = every list to its end; 1 = the first window, the ordered
  // length goes to sorted[tile] (and open[tile] = 0); 2 = the lists whose compositing ran out of ordered entries (open[tile]
  // != 0), to their end; 3 = every list that is not ordered to its end (lograst_finish_lists).  2 and 3 run the list again
  // from its keys (the bucket path never moves them) and skip the windows that are in place.
  uint32_t* const sorted = state + lr_sorted_off(tiles) + tile;
  uint32_t* const open = sorted + tiles;
  uint32_t in_place = 0u;                                   // leading positions a previous pass left in final order
  if (lazy >= 2) {
    if (L <= LR_LONG_LIST || !state[LR_HDR_LAZY]) return;
    in_place = *sorted;
    if (lazy == 2 ? *open == 0u : (*open != 0u || in_place >= L)) return;
    __syncthreads();                                        // (everybody has read the word that thread 0 may rewrite at the end)
  }
  const uint64_t* k = keys + beg;
  uint16_t* rk = reinterpret_cast<uint16_t*>(ranks) + beg;   // one 16-bit bucket id per key (nb <= 4096)
  uint32_t* pl = plist + beg;
  // Clustered depths (a bucket above LR_BUCKET_MAX keys after both maps) or LOGRAST_BUCKET_SORT=0: the network, by this
  // workgroup, in place on the tile's keys (the whole dynamic LDS block as its staging area), then the ids.
  auto network = [&]() {
    __syncthreads();
    lr_wg_hybrid_sort<1024>(keys + beg, L, reinterpret_cast<uint64_t*>(lcnt), tid);
    for (uint32_t i = tid; i < L; i += 1024u) pl[i] = (uint32_t)k[i];
    if (lazy && lazy != 2 && tid == 0 && L > LR_LONG_LIST) { *sorted = L; if (lazy == 1) *open = 0u; }
  };
  if (network_only) {                                       // (lists up to one block were sorted by lr_sort_rb_kernel)
    if (L > LR_SORT_BLOCK) network();
    return;
  }
  // up to 4096 keys: the whole list in LDS, same code as the small lists.  (Up to 8192 the keys could sit in registers --
  // that was a kernel of its own, 145 VGPRs -- but not at the 64 this kernel is held to: they stream like the longer ones.)
  if (L <= LR_LONG_LIST) { lr_bucket_tile<1024, 4>(keys, plist, beg, L, equalize, reinterpret_cast<uint64_t*>(lcnt)); return; }
#if defined(LR_EXPERIMENTS) && defined(LR_LONG_TICKS)   // phase timing experiment (-DLR_EXPERIMENTS -DLR_LONG_TICKS): wall_clock64 per phase, printed by three workgroups
  uint64_t tk[16]; int tn = 0;
#define LR_TICK() do { if (tn < 16) tk[tn++] = wall_clock64(); } while (0)
#else
#define LR_TICK() do { } while (0)
#endif
  LR_TICK();
  if (lazy == 1 && tid == 0) state[LR_HDR_LAZY] = 1u;       // (every streamed list's workgroup stores the same word)
  uint32_t P2 = 8;
  while (P2 < L) P2 <<= 1;
  const uint32_t nb = min((uint32_t)LR_LONG_NB, P2 >> 1);
  if (tid == 0) { sh_min = 0xffffffffu; sh_max = 0u; sh_maxcnt = 0u; }
  if (tid < LR_CELLS) cellcnt[tid] = 0u;
  for (uint32_t b = tid; b < nb; b += 1024) lcnt[b] = 0u;
  uint32_t dmin = 0xffffffffu, dmax = 0u;
  // (the streaming loops are unrolled by hand: LR_LONG_UNR independent loads in flight per thread -- every pass is a
  // latency chain per workgroup, 1024 threads x 8 loads x 8 B = 64 KB in flight)
  // Depth range from a sample: Ls keys in 64-key pieces (one coalesced 512-byte read per wave) spread evenly over the
  // list -- the keys arrive in Gaussian order, which is no particular depth order for a random cloud but IS one for a
  // level-of-detail selection (coarse levels first), so a prefix would be a biased sample.  The range only has to
  // spread the keys over the buckets: keys outside it clamp into the first / last bucket, and a bucket that overflows
  // sends the tile to the second attempt / the fallback as before.
  const uint32_t Ls = min(L, max(L >> 4, 2048u)) & ~63u;    // (L > LR_SORT_BLOCK here: at least 2048)
  const uint32_t npieces = Ls >> 6;
  auto sample_at = [&](uint32_t j) -> uint32_t {            // list position of sample j
    return (uint32_t)(((uint64_t)(j >> 6) * L) / npieces) + (j & 63u);
  };
  for (uint32_t i = tid; i < Ls; i += LR_LONG_UNR * 1024u) {
    uint64_t kk[LR_LONG_UNR];
#pragma unroll
    for (int u = 0; u < LR_LONG_UNR; u++) kk[u] = (i + u * 1024u < Ls) ? k[sample_at(i + u * 1024u)] : 0ull;
#pragma unroll
    for (int u = 0; u < LR_LONG_UNR; u++)
      if (i + u * 1024u < Ls) { const uint32_t d = (uint32_t)(kk[u] >> 32); dmin = min(dmin, d); dmax = max(dmax, d); }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, d));
    dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, d));
  }
  __syncthreads();
  if ((tid & 63u) == 0u) { atomicMin(&sh_min, dmin); atomicMax(&sh_max, dmax); }
  __syncthreads();
  LR_TICK();
  // (the sampled range widened by 1 % on either side: the few keys beyond the sample's extremes lie just outside it and
  // get buckets of their own there; clamped into the first / last bucket they overflow it -- measured: every eighth
  // tile of the 30 M-Gaussian workload then took the network fallback)
  const float smin = __uint_as_float(sh_min), srange = __uint_as_float(sh_max) - smin;
  LrDepthMap dmap;
  dmap.fmin = smin - 0.01f * srange;
  dmap.ncells = min((uint32_t)LR_CELLS, nb >> 1);
  dmap.cscale = (float)dmap.ncells / (1.02f * srange);
  dmap.table = celltab;
  auto bucket_of = [&](uint64_t key) -> uint32_t { return lr_depth_bucket(dmap, (uint32_t)(key >> 32)); };
  // first the plain linear map; if a bucket overflows, once more with the map equalised over the first 8192 keys
  // (cache-resident: just read for the range) -- see lr_sort_bucket_kernel
  const uint32_t per = (nb + 1023u) / 1024u;               // <= LR_LONG_NB / 1024
  uint32_t cown[LR_LONG_NB / 1024];
  uint32_t local = 0, inc = 0;
  for (int attempt = 0;; attempt++) {
    const bool eq = attempt == 1;
    const uint32_t Lh = min(Ls, 8192u);
    if (eq)
      for (uint32_t i = tid; i < Lh; i += 1024u) { float fr; atomicAdd(&cellcnt[lr_depth_cell(dmap, (uint32_t)(k[sample_at(i)] >> 32), fr)], 1u); }
    __syncthreads();
    lr_depth_map_build(celltab, cellcnt, dmap.ncells, Lh, nb, eq);
    __syncthreads();
    for (uint32_t i = tid; i < L; i += LR_LONG_UNR * 1024u) {  // bucket of every key + the bucket sizes
      uint64_t kk[LR_LONG_UNR];
#pragma unroll
      for (int u = 0; u < LR_LONG_UNR; u++) kk[u] = (i + u * 1024u < L) ? k[i + u * 1024u] : 0ull;
      uint32_t cd[LR_LONG_UNR];
#pragma unroll
      for (int u = 0; u < LR_LONG_UNR; u++) {                  // the bucket id is all a window pass needs to skip a key
        cd[u] = bucket_of(kk[u]);
        if (i + u * 1024u < L) atomicAdd(&lcnt[cd[u]], 1u);   // (count only: no returning atomic, nothing waits for it)
      }
#pragma unroll
      for (int u = 0; u < LR_LONG_UNR; u++)
        if (i + u * 1024u < L) rk[i + u * 1024u] = (uint16_t)cd[u];
    }
    __syncthreads();
    LR_TICK();
    // exclusive scan of the nb counts: thread t owns counters [t * per, (t + 1) * per); every wave scans the 16 wave totals
    uint32_t lmax = 0;
    local = 0;
#pragma unroll
    for (uint32_t q = 0; q < LR_LONG_NB / 1024; q++) {
      const uint32_t b = tid * per + q;
      cown[q] = (q < per && b < nb) ? lcnt[b] : 0u;
      local += cown[q]; lmax = max(lmax, cown[q]);
    }
    inc = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(inc, d);
      if ((int)(tid & 63u) >= d) inc += up;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, d));
    if ((tid & 63u) == 63u) { wave_tot[tid >> 6] = inc; atomicMax(&sh_maxcnt, lmax); }
    __syncthreads();
    if (sh_maxcnt <= LR_BUCKET_MAX || eq || !equalize) break;
    __syncthreads();                                          // every thread has read sh_maxcnt and its counts
    if (tid == 0) sh_maxcnt = 0u;
    for (uint32_t b = tid; b < nb; b += 1024) lcnt[b] = 0u;
  }
  if (sh_maxcnt > LR_BUCKET_MAX) { network(); return; }
  {
    const uint32_t wt = (tid & 63u) < 16u ? wave_tot[tid & 63u] : 0u;
    uint32_t winc = wt;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const uint32_t up = __shfl_up(winc, d);
      if ((int)(tid & 63u) >= d) winc += up;
    }
    uint32_t run = (uint32_t)__shfl((int)(winc - wt), (int)(tid >> 6)) + inc - local;   // waves in front of mine + lanes in front of me
#pragma unroll
    for (uint32_t q = 0; q < LR_LONG_NB / 1024; q++) {
      const uint32_t b = tid * per + q;
      if (q < per && b < nb) { lcnt[b] = run; run += cown[q]; }   // lcnt[b] = first list position of bucket b
    }
  }
  __syncthreads();
  LR_TICK();
  // windows of whole buckets: [b0, b1) with start(b1) - start(b0) <= LR_LONG_WIN (a bucket holds <= LR_BUCKET_MAX keys)
  uint32_t b0 = 0;
  while (b0 < nb) {
    const uint32_t w0 = lcnt[b0];
    uint32_t lo_b = b0 + 1u, hi_b = nb;                    // largest b1 in (b0, nb] with start(b1) <= w0 + WIN
    while (lo_b < hi_b) {
      const uint32_t mid = (lo_b + hi_b + 1u) >> 1;
      const uint32_t st_mid = mid < nb ? lcnt[mid] : L;
      if (st_mid <= w0 + (uint32_t)LR_LONG_WIN) lo_b = mid; else hi_b = mid - 1u;
    }
    const uint32_t b1 = lo_b;                              // window = list positions [start(b0), start(b1))
    if (lazy >= 2) {   // windows an earlier pass left in place are skipped (same cuts: same keys, same map)
      const uint32_t wend = b1 < nb ? lcnt[b1] : L;
      __syncthreads();   // everybody has read it: a thread that skips goes straight on to the NEXT window, whose scatter bumps lcnt[b1]
      if (wend <= in_place) { b0 = b1; continue; }
    }
    for (uint32_t i = tid; i < L; i += LR_LONG_UNR * 1024u) {  // 2 B per key; the 8-byte key only if it lands in this window
      uint32_t cc[LR_LONG_UNR];
#pragma unroll
      for (int u = 0; u < LR_LONG_UNR; u++) cc[u] = (i + u * 1024u < L) ? (uint32_t)rk[i + u * 1024u] : 0xffffu;
      uint64_t kv[LR_LONG_UNR];
#pragma unroll
      for (int u = 0; u < LR_LONG_UNR; u++) {                // all key loads requested before the first is used
        const uint32_t b = cc[u];                            // 0xffff (past the end) >= nb
        kv[u] = (b >= b0 && b < b1) ? k[i + u * 1024u] : 0ull;
      }
      // the key's slot: the bucket's start counter doubles as its fill cursor (any order inside the bucket will do, the
      // emit below orders it by the full key); after the pass lcnt[b] = end of bucket b = start of bucket b + 1
      uint32_t ps[LR_LONG_UNR];
#pragma unroll
      for (int u = 0; u < LR_LONG_UNR; u++) {
        const uint32_t b = cc[u];
        ps[u] = (b >= b0 && b < b1) ? atomicAdd(&lcnt[b], 1u) : 0u;
      }
#pragma unroll
      for (int u = 0; u < LR_LONG_UNR; u++) {
        const uint32_t b = cc[u];
        if (b >= b0 && b < b1) win[ps[u] - w0] = kv[u];
      }
    }
    __syncthreads();
    LR_TICK();
    for (uint32_t bb = b0; bb < b1; bb += 1024u) {           // (uniform trip count: the emit needs whole waves)
      const uint32_t b = bb + tid;
      const bool valid = b < b1;
      // the ordered ids go back into the (now consumed) window slots of their bucket, low word of the slot at the final
      // position: the copy-out below then writes the list in full lines instead of 4-byte pieces 20 bytes apart
      lr_emit_bucket([&](uint32_t pos) -> uint64_t { return win[pos - w0]; },   // win[] holds positions [w0, w1)
                     valid ? (b == b0 ? w0 : lcnt[b - 1u]) : 0u, valid ? lcnt[b] : 0u, valid,
                     reinterpret_cast<uint32_t*>(win) - 2 * (size_t)w0, 2u);
    }
    __syncthreads();
    {
      const uint32_t w1 = b1 < nb ? lcnt[b1] : L;
      const uint32_t* ids = reinterpret_cast<const uint32_t*>(win);
      for (uint32_t p = w0 + tid; p < w1; p += 1024u) pl[p] = ids[2u * (p - w0)];
    }
    __syncthreads();
    LR_TICK();
    if (lazy == 1 && b1 < nb) {                              // the first window is what a view walks; the rest on demand
      if (tid == 0) { *sorted = lcnt[b1]; *open = 0u; }      // (bucket b1 is untouched: lcnt[b1] is still its first position)
      return;
    }
    b0 = b1;
  }
  if (lazy && lazy != 2 && tid == 0) { *sorted = L; if (lazy == 1) *open = 0u; }   // (2: sorted[] stays where the parked waves resume)

#if defined(LR_EXPERIMENTS) && defined(LR_LONG_TICKS)
  if (tid == 0 && (blk == 0 || blk == 700 || blk == 2000)) {
    printf("longsort blk %u L %u nb %u ticks(10ns):", blk, L, nb);
    for (int q = 1; q < tn; q++) printf(" %llu", (unsigned long long)(tk[q] - tk[q - 1]));
    printf("\n");
  }
#endif
#undef LR_TICK
}
// First pass (lazy 0 / 1): one workgroup per list, handed out longest first by the dispatcher.  The passes over the tails
// (lazy 2 / 3) run a small resident grid that loops over the lists: the per-view launch of mode 2 finds nothing to do in
// nearly every view (LR_HDR_OPEN: no wave parked), and 512 workgroups that return after two loads cost a tenth of 8160.
// (Two instantiations: wrapped in the loop the list code keeps its arguments live across iterations and spills -- 148 bytes
// of scratch per lane against 8 -- which the first pass, the one every view pays for, must not inherit.)
template <bool REST>
__global__ void __launch_bounds__(1024, 8)   // two workgroups per CU: 64 VGPRs (at 77 the kernel ran one per CU: 0.62 -> 0.86 ms at 30 M)
lr_sort_long_kernel(uint32_t* __restrict__ state, uint32_t tiles, uint64_t* __restrict__ keys,
                    uint32_t* __restrict__ ranks, uint32_t* __restrict__ plist, uint32_t capacity, int equalize,
                    int network_only, int lazy, uint32_t nblk) {
  if (lr_bail(state, capacity)) return;
  if (!REST) {
    lr_sort_long_list(state, tiles, keys, ranks, plist, equalize, network_only, lazy, blockIdx.x);
    return;
  }
  if (lazy == 2 && !state[LR_HDR_OPEN]) return;
  for (uint32_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    lr_sort_long_list(state, tiles, keys, ranks, plist, equalize, network_only, lazy, blk);
    __syncthreads();                                        // (the next list reuses the LDS)
  }
}

static inline size_t lr_sort_lds_bytes(uint32_t cap) { return sizeof(uint64_t) * (size_t)(cap + (cap >> 3)); }
static inline size_t lr_long_lds_bytes() {   // (also >= lr_bucket_lds_bytes(4096) = 73728)
  return sizeof(uint32_t) * LR_LONG_NB + LR_LONG_WIN_BYTES + sizeof(uint64_t) * LR_BUCKET_MAX;
}
static_assert(sizeof(uint32_t) * LR_LONG_NB + LR_LONG_WIN_BYTES + sizeof(uint64_t) * LR_BUCKET_MAX >=
              sizeof(uint64_t) * (LR_SORT_BLOCK + (LR_SORT_BLOCK >> 3)), "the fallback network stages one block in the same LDS");

// Size classes: the LDS footprint (9 B/key with padding) sets how many workgroups a CU can hold, so small
// lists must not pay for the largest class.
#define LR_SORT_CAP0 512             // 64 threads (one wave), 4.5 KB
#define LR_SORT_CAP1 2048            // 256 threads, 18 KB
#define LR_SORT_CAP2 LR_SORT_BLOCK   // 256 threads, 72 KB (dynamic LDS beyond the 64 KB static limit)

// max_len: upper bound on the longest tile list known to the HOST (exact count from stage 1, a hint in sync-free
// operation, or 0 = unknown -> assume `capacity`).  It only decides how many multi-block levels are launched.
// -> the lazy mode that is really in effect (experiment builds with LOGRAST_BUCKET_SORT=0 order every list to its end: the
// compositing passes must then not read sorted[] / open[], which still hold dead fill cursors -- round-5 advisory).
int lr_launch_sort(uint32_t* state, uint32_t tiles, uint64_t* keys, uint32_t* plist, uint32_t capacity,
                   uint32_t max_len, int lazy, hipStream_t s) {
  if (tiles == 0) return 0;
  static bool attr_set = false;
  if (!attr_set) {
    const int big = (int)lr_sort_lds_bytes(LR_SORT_BLOCK);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lr_sort_rb_kernel<256>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lr_sort_long_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lr_long_lds_bytes());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lr_sort_long_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lr_long_lds_bytes());
    attr_set = true;
  }
  if (max_len == 0 || max_len > capacity) max_len = capacity;
  // LOGRAST_BUCKET_SORT=0: bitonic network only (the reference implementation of the same total order)
  static const int bucket = LR_EXPERIMENT_INT("LOGRAST_BUCKET_SORT", 1);   // experiment builds: 0 = bitonic network only
  static const int equalize = LR_EXPERIMENT_INT("LOGRAST_EQUALIZE", 1);   // 0: plain linear depth -> bucket map (experiments)
  // Two launches: one 256-thread workgroup per tile for the lists of up to 1024 keys, and one 1024-thread workgroup
  // (78 KB of LDS, two per CU) per list above that, walking the longest-first order -- LDS-resident up to 4096 keys,
  // keys in registers up to 8192, streamed from memory beyond.  (They used to be four launches by size class: at 30 M
  // Gaussians, where every list is long, the three that found nothing to do cost 100 us of idle workgroups per view.)
  if (bucket) {
    lr_prof_begin(LRK_SORT_SMALL, s);
    hipLaunchKernelGGL(lr_sort_small_kernel, dim3(tiles), dim3(256), lr_bucket_lds_bytes(1024), s, state, tiles, keys,
                       plist, capacity, equalize);
    lr_prof_end(LRK_SORT_SMALL, s);
  } else {
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
  }
  if (bucket ? max_len > 1024u : max_len > LR_SORT_BLOCK) {
    lr_prof_begin(LRK_SORT_HUGE, s);
    const uint32_t nblk = min(tiles, capacity / 1024u + 1u);
    hipLaunchKernelGGL(lr_sort_long_kernel<false>, dim3(nblk), dim3(1024), lr_long_lds_bytes(), s,
                       state, tiles, keys, reinterpret_cast<uint32_t*>(keys + capacity), plist, capacity, equalize,
                       bucket ? 0 : 1, (bucket && lazy) ? 1 : 0, nblk);
    lr_prof_end(LRK_SORT_HUGE, s);
  }
  return (bucket && lazy) ? 1 : 0;
}

// out[t] = leading positions of tile t's list that are in final order (lograst_ordered_lengths)
__global__ void __launch_bounds__(256)
lr_ordered_lengths_kernel(const uint32_t* __restrict__ state, uint32_t tiles, uint32_t* __restrict__ out) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= tiles) return;
  const uint32_t* offsets = state + lr_offsets_off(tiles);
  const uint32_t L = offsets[t + 1] - offsets[t];
  const bool lazy = state[LR_HDR_LAZY] != 0u && L > LR_LONG_LIST;
  const uint32_t* sorted = state + lr_sorted_off(tiles);
  out[t] = (lazy && sorted[tiles + t] == 0u) ? min(sorted[t], L) : L;   // (open[t] != 0: the second pass ordered it to its end)
}
void lr_launch_ordered_lengths(const uint32_t* state, uint32_t tiles, uint32_t* out, hipStream_t s) {
  if (tiles == 0) return;
  hipLaunchKernelGGL(lr_ordered_lengths_kernel, dim3((tiles + 255u) / 256u), dim3(256), 0, s, state, tiles, out);
}

// The rest of the lists that lr_launch_sort(lazy = 1) left at their first window: mode 2 = those whose compositing asked for
// it (open[tile] != 0), mode 3 = all of them (lograst_finish_lists).  Same grid as the first pass; a workgroup whose list
// needs nothing returns after two loads.
void lr_launch_sort_rest(uint32_t* state, uint32_t tiles, uint64_t* keys, uint32_t* plist, uint32_t capacity,
                         uint32_t max_len, int mode, hipStream_t s) {
  if (tiles == 0) return;
  if (max_len == 0 || max_len > capacity) max_len = capacity;
  if (max_len <= LR_LONG_LIST) return;
  static const int equalize = LR_EXPERIMENT_INT("LOGRAST_EQUALIZE", 1);
  // (every tile in front of a streamed list in order[] holds at least LR_LONG_LIST keys itself: lr_scan_kernel's buckets)
  static bool attr_set = false;   // (lograst_finish_lists may be the first caller in a process that only inspects buffers)
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lr_sort_long_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lr_long_lds_bytes());
    attr_set = true;
  }
  const uint32_t nblk = min(tiles, capacity / (uint32_t)LR_LONG_LIST + 1u);
  hipLaunchKernelGGL(lr_sort_long_kernel<true>, dim3(min(nblk, 512u)), dim3(1024), lr_long_lds_bytes(), s, state, tiles, keys,
                     reinterpret_cast<uint32_t*>(keys + capacity), plist, capacity, equalize, 0, mode, nblk);
}

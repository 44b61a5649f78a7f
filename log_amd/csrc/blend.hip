// blend.hip -- A5/A7/A8 forward compositing and A6 backward reverse walk.
//
// CDNA4 mapping (not the 16x16-threads-per-tile CUDA shape with shared-memory staging and block barriers):
//   * a 256-thread workgroup owns one 16x16 tile, but its four 64-lane waves never synchronise: wave q owns
//     the 8x8 pixel quadrant q (one pixel per lane) and walks the tile's depth-sorted list on its own;
//   * per 64 list entries a wave gathers 64 projected records into VGPRs (ids read coalesced from the
//     sorted list; the four waves of a tile hit the same lines in their CU's L1), computes for each a
//     conservative alpha-support box, and ballots which entries can touch ITS quadrant at all;
//   * only those entries are visited (s_ff1 over the ballot mask): one v_readlane fetches the entry's id and its
//     record then comes through the SCALAR cache (uniform address -> s_load_dwordx8), so the current Gaussian
//     lives in SGPRs -- no VALU broadcast, no LDS traffic, no barrier, no VGPRs -- culling a (Gaussian,
//     quadrant) pair costs nothing, and every wave stops exactly when its own 64 pixels are saturated.
// The support box only skips pixels whose alpha is provably below the 1/255 floor (1% margin on the
// exponent, disabled for ill-conditioned conics), so results are identical to visiting every entry; the
// oracle does not cull and the bit-exact parity tests check exactly that.
//
// Backward: same mapping; each (wave, Gaussian) gradient is reduced across the wave with a packed
// v_permlane32_swap / v_permlane16_swap + DPP row-rotate tree (28 VALU ops for 9 sums) and committed with
// 3 atomic instructions -- not 9 atomics per (pixel, Gaussian) as in the third-party kernel.
#include "common.hpp"
// "These prefetched values have landed": an empty asm that reads and writes them -- the compiler has to wait for their loads
// right here.  (A bare __builtin_amdgcn_s_waitcnt in front of a loop is hoisted above the loads it was meant for.)
#define LR_LANDED(q0, q1, c, i)                                                                                  \
  asm volatile("" : "+v"((q0).x), "+v"((q0).y), "+v"((q0).z), "+v"((q0).w), "+v"((q1).x), "+v"((q1).y), "+v"((q1).z), \
               "+v"((q1).w), "+v"(c), "+v"(i))
// Occupancy hints (waves per SIMD the register allocator aims for), per kernel: compile-time so that A/B builds
// (`python -m log_amd.build <variant> -DLR_OCC_BWD_ROWS_WAVES=5`) can measure them; empty = the allocator's own choice.
#define LR_OCC_ATTR(n) __attribute__((amdgpu_waves_per_eu(n)))
#ifdef LR_OCC_BWD_ROWS_WAVES
#define LR_OCC_BWD_ROWS LR_OCC_ATTR(LR_OCC_BWD_ROWS_WAVES)
#else
#define LR_OCC_BWD_ROWS
#endif
#ifndef LR_OCC_FWD_ROWS_WAVES
#define LR_OCC_FWD_ROWS_WAVES 5   // round 6: with the unconditional prefetch loads the allocator's own choice is 102 VGPRs (4 waves); 96 fit without a spill
#endif
#define LR_OCC_FWD_ROWS LR_OCC_ATTR(LR_OCC_FWD_ROWS_WAVES)
#ifdef LR_OCC_BWD_WAVES
#define LR_OCC_BWD LR_OCC_ATTR(LR_OCC_BWD_WAVES)
#else
#define LR_OCC_BWD
#endif
#ifdef LR_OCC_FWD_WAVES
#define LR_OCC_FWD LR_OCC_ATTR(LR_OCC_FWD_WAVES)
#else
#define LR_OCC_FWD
#endif


// blockIdx -> tile.  Workgroup b is observed to run on XCD b % 8 (speed only, never correctness).
//   mode 0: identity -- consecutive tiles round-robin over the XCDs: best balance, no L2 sharing;
//   mode 1: XCD-banded -- XCD k gets the contiguous tile range [k*nper,(k+1)*nper): best L2 sharing, but a
//           scene that fills only part of the screen leaves whole XCDs idle;
//   mode 2: 4x4-tile super-blocks dealt round-robin to the XCDs: neighbouring tiles (which share Gaussians)
//           share an L2, and every XCD still gets a slice of every screen region.
//   mode 3: longest list first (order[] written by the scan kernel): the dispatcher hands out workgroups in
//           index order, so the serial walks of the longest lists start first instead of forming the tail.
LR_DEV uint32_t lr_tile_of_block(uint32_t b, uint32_t tiles, int gx, int gy, int mode,
                                 const uint32_t* __restrict__ state) {
  if (mode == 3) return b < tiles ? state[lr_order_off(tiles) + b] : 0xffffffffu;
  if (mode == 1) {
    uint32_t nper = (tiles + 7u) >> 3;
    uint32_t t = (b & 7u) * nper + (b >> 3);
    return t < tiles ? t : 0xffffffffu;
  }
  if (mode == 2) {
    uint32_t nsbx = ((uint32_t)gx + 3u) >> 2, nsby = ((uint32_t)gy + 3u) >> 2;
    uint32_t r = b >> 3, sb = (r >> 4) * 8u + (b & 7u), local = r & 15u;
    if (sb >= nsbx * nsby) return 0xffffffffu;
    uint32_t tx = (sb % nsbx) * 4u + (local & 3u), ty = (sb / nsbx) * 4u + (local >> 2);
    if (tx >= (uint32_t)gx || ty >= (uint32_t)gy) return 0xffffffffu;
    return ty * (uint32_t)gx + tx;
  }
  return b < tiles ? b : 0xffffffffu;
}
static inline __host__ __device__ uint32_t lr_blend_grid(uint32_t tiles, int gx, int gy, int mode) {
  if (mode == 1) return ((tiles + 7u) / 8u) * 8u;
  if (mode == 2) {
    uint32_t nsb = (((uint32_t)gx + 3u) >> 2) * (((uint32_t)gy + 3u) >> 2);
    return ((nsb + 7u) / 8u) * 8u * 16u;
  }
  return tiles;
}

// Conservative test: can the Gaussian (record q0,q1) reach alpha >= 1/255 anywhere in the pixel box
// [x0,x1]x[y0,y1]?  The alpha floor means power >= -tau, tau = ln(255*opacity); the level set
// {d : 0.5 d^T Q d <= tau'} has the axis-aligned half extents sqrt(2 tau' cov_xx), sqrt(2 tau' cov_yy).
// tau' = 1.01 tau + 0.01 absorbs the fp32 evaluation error of `power`; if that error could exceed the
// margin (ill-conditioned conic, non-finite values) the answer is "yes" (never cull).  Two stages: the cheap
// box-vs-box test, then the exact ellipse-vs-box test (the box of an elongated, tilted ellipse is mostly empty).
LR_DEV bool lr_support_hits(const float4 g0, const float2 g1, float x0, float x1, float y0, float y1) {
  const float A = g0.z, B = g0.w, C = g1.x, op = g1.y;
  const float det = A * C - B * B;
  const float tau = lr_fma(__logf(255.f * op), 1.01f, 0.01f);
  if (!(op >= 1.0f / 512.0f)) return !(op < 1.0f / 512.0f);  // tiny opacity never reaches the floor; NaN -> keep
  const float inv = 1.f / det;
  const float ex2 = 2.f * tau * C * inv, ey2 = 2.f * tau * A * inv;  // squared half extents
  const float mag = (fabsf(A) + fabsf(B) + fabsf(C)) * (ex2 + ey2);  // bound on the terms of `power` in the box
  const bool safe = (det > 0.f) && (ex2 >= 0.f) && (ey2 >= 0.f) && (mag * 1.0e-6f < 0.005f * tau) && (mag < 1.0e30f);
  if (!safe) return true;
  const float ex = sqrtf(ex2) + 0.01f, ey = sqrtf(ey2) + 0.01f;
  if (!((g0.x + ex >= x0) && (g0.x - ex <= x1) && (g0.y + ey >= y0) && (g0.y - ey <= y1))) return false;
  // The axis-aligned box of the ellipse overlaps; now the ellipse itself: min over the pixel box of
  // q(d) = 0.5 d^T Q d.  If the centre is inside the (0.01-inflated) box the minimum is 0; otherwise it lies on
  // one of the four edges, where q is a 1-D parabola whose clamped vertex is closed-form.
  const float dx0 = (x0 - 0.01f) - g0.x, dx1 = (x1 + 0.01f) - g0.x;
  const float dy0 = (y0 - 0.01f) - g0.y, dy1 = (y1 + 0.01f) - g0.y;
  if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return true;
  const float iA = 1.f / A, iC = 1.f / C;
  float best = 3.0e38f;
#pragma unroll
  for (int e = 0; e < 2; e++) {
    const float dx = e ? dx1 : dx0;
    const float dy = fminf(dy1, fmaxf(dy0, -B * dx * iC));
    best = fminf(best, 0.5f * (A * dx * dx + C * dy * dy) + B * dx * dy);
    const float ey_ = e ? dy1 : dy0;
    const float ex_ = fminf(dx1, fmaxf(dx0, -B * ey_ * iA));
    best = fminf(best, 0.5f * (A * ex_ * ex_ + C * ey_ * ey_) + B * ex_ * ey_);
  }
  return !(best > tau);
}

// Lazily ordered lists (common.hpp: sorted[] / open[]; sort.hip).  lazy = 1: the first compositing pass -- a streamed list
// is walked over its ordered part only (`end` comes back clamped); a wave that gets there with a pixel still open parks
// its pixels' state in the outputs (lr_lazy_park) and sets its bit in open[tile].  lazy = 2: the second pass, after
// lr_launch_sort_rest ordered those lists to their end -- only the waves that parked run (false = nothing to do for this
// one): they pick their state up again and go on at list position `first`, entry for entry what an uninterrupted walk does.
LR_DEV bool lr_lazy_range(const uint32_t* sorted, uint32_t tiles, int lazy, uint32_t tile, int wave, uint32_t beg,
                          uint32_t& end, uint32_t& first, bool& clamped) {
  first = 0u;
  clamped = false;
  if (!lazy) return true;
  if (end - beg <= LR_LONG_LIST) return lazy == 1;
  // (down to a whole number of 64-entry chunks: the chunks of both passes are then the chunks of an uninterrupted walk,
  // which is what the hit masks handed to the reverse walk are indexed by; the entries in between are simply walked later)
  const uint32_t ordered = sorted[tile] & ~63u;
  if (lazy == 1) {
    if (ordered < end - beg) { end = beg + ordered; clamped = true; }
    return true;
  }
  if (!((sorted[tiles + tile] >> wave) & 1u)) return false;
  first = ordered;
  return true;
}
// A pixel's compositing state between the two passes lives in its own outputs: T in final_T (negative = the pixel has
// stopped: T itself never drops below 1e-4), the colour sums WITHOUT the background term in the image, the last contributor
// and the fork maps' running maximum where they will end up anyway.
template <bool EXTRAS>
LR_DEV void lr_lazy_park(const LrView& v, size_t pix, bool done, float T, float C0, float C1, float C2, int last, int wid,
                         float wmax, float* image, float* final_T, int* n_contrib, int* pid, float* pwp) {
  const size_t plane = (size_t)v.W * v.H;
  final_T[pix] = done ? -T : T;
  n_contrib[pix] = last;
  image[pix] = C0; image[plane + pix] = C1; image[2 * plane + pix] = C2;
  if (EXTRAS) { pid[pix] = wid; pwp[pix] = wmax; }
}
template <bool EXTRAS>
LR_DEV void lr_lazy_resume(const LrView& v, size_t pix, bool& done, float& T, float& C0, float& C1, float& C2, int& last,
                           int& wid, float& wmax, const float* image, const float* final_T, const int* n_contrib,
                           const int* pid, const float* pwp) {
  const size_t plane = (size_t)v.W * v.H;
  const float t = final_T[pix];
  done = t < 0.f;
  T = fabsf(t);
  last = n_contrib[pix];
  C0 = image[pix]; C1 = image[plane + pix]; C2 = image[2 * plane + pix];
  if (EXTRAS) { wid = pid[pix]; wmax = pwp[pix]; }
}

// ---- hit masks: the forward's support decisions handed to the reverse walk (round 6) ------------------------------------
// Per 64-entry chunk of a tile's list a compositing wave ballots which entries can reach the alpha floor inside its
// quadrant (quadrant form: one 64-bit mask) or inside each of its four 4x4 blocks (row-split form: four masks).  The reverse
// walk visits the same (wave, chunk) pairs and used to run the same tests again -- lr_support_prepare + two lr_support_box2
// per entry, ~190 VALU: a third of lr_blend_bwd_rows_kernel -- and to gather all 64 records of a chunk to run them on.  When
// the caller provides lograst_view.hit_masks the forward stores its ballots (8 B per wave and chunk, 4 x 8 B in the row-split
// form; only the chunks it really walked: a few per cent of a long list) and the reverse walk, now walking the list in the
// forward's chunks (aligned to multiples of 64 instead of to its deepest contributor), loads them through the scalar cache,
// gathers only the records whose bit is set and runs no support test.  Same decisions, so the same visits and the same sums.
// Slot of (tile, chunk c): (offsets[tile] >> 6) + tile + c -- disjoint for all tiles (floor(L / 64) + 1 >= ceil(L / 64)),
// at most capacity / 64 + tiles + 1 slots; a slot holds 16 words (row-split: [wave][block]) or 4 (quadrant: [wave]).
// Which form wrote them: lograst_view.hit_mask_form of the backward's view (the caller knows what its forward launched:
// lograst_forward_form); a reverse walk of the other form ignores the buffer and runs the tests as before.  The forward also
// leaves the form in header word LR_HDR_MASKS (diagnostics).
#define LR_MBUF_CHUNKS 64u   // row-split forward: chunks of hit masks a wave collects in LDS between two bursts of stores
#define LR_MASK_FORM_ROWS 1u
#define LR_MASK_FORM_QUAD 2u
LR_DEV size_t lr_mask_slot(uint32_t list_begin, uint32_t tile) { return (size_t)(list_begin >> 6) + tile; }
// bits of a REVERSED chunk mask (bit j = list position hi - 1 - j) whose position lies in front of `limit`
LR_DEV uint64_t lr_mask_before(uint64_t m, int hi, int limit) {
  const int sh = hi - limit;                                 // positions hi - 1 - j < limit  <=>  j >= sh
  return sh <= 0 ? m : (sh >= 64 ? 0ull : (m & (~0ull << sh)));
}

// ---- the forward's per-Gaussian outputs, committed once per chunk (round 6) ---------------------------------------------
// A contributing visit used to end in memory operations of its own: an atomicMax on point_weight[g] and -- in a training
// forward on a large input -- the clearing of g's accumulator row.  The experiment build showed what they cost
// (profiles/r06_fwd_memops.jsonl, 30 M Gaussians, row-split form): the kernel takes 572-595 us with them and 407-417 without
// (opacity = rand: 1098 / 713) -- and leaving out EITHER kind alone already gives 437 / 441: not their throughput, their
// place.  gfx9 counts loads, stores and atomics in one counter that retires IN ORDER, so the wait for the next chunk's
// prefetched records at the end of every chunk also waited for the last atomic / store the pass loop had just issued -- a
// memory-side round trip of its own, once per chunk and wave.  Now a visit raises a per-wave LDS maximum of its entry
// (ds_max_u32: one lane), and the chunk's maxima and row clears leave in ONE burst at the top of the NEXT chunk, in front of
// that chunk's prefetch loads: nothing the chunk's final wait covers is younger than the loads it is for, and the burst
// has a whole chunk of passes to complete.  Same values (a maximum of the same numbers; the same rows cleared).
template <bool EXTRAS>
LR_DEV void lr_fwd_commit_chunk(uint32_t* wmx, int lane, uint32_t id_prev, float* __restrict__ pw, float4* __restrict__ zero_rows) {
  if (!EXTRAS) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const uint32_t m = (id_prev != 0xffffffffu) ? wmx[lane] : 0u;
  if (m != 0u) atomicMax(reinterpret_cast<unsigned int*>(pw) + id_prev, m);
  // training forward: this Gaussian contributes, so the reverse walk will add to its 64-byte accumulator row -- clear it
  // (every wave that meets the Gaussian stores the same zeros; rows of Gaussians nobody meets are never read).  Four lanes
  // per row, one quarter each: every store instruction writes up to sixteen COMPLETE lines (a lane clearing its own row in
  // four instructions is four partial writes per line at the L2).
  if (zero_rows) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int e = 16 * k + (lane >> 2);
      const uint32_t me = (uint32_t)__shfl((int)m, e), ide = (uint32_t)__shfl((int)id_prev, e);
      if (me != 0u) zero_rows[4 * (size_t)ide + (lane & 3)] = float4{0.f, 0.f, 0.f, 0.f};
    }
  }
  wmx[lane] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <bool EXTRAS>
__global__ void __launch_bounds__(256) LR_OCC_FWD
lr_blend_fwd_kernel(LrView v, const float4* __restrict__ geom, const uint32_t* __restrict__ state,
                    uint32_t tiles, const uint32_t* __restrict__ plist, uint32_t capacity,
                    float* __restrict__ image, float* __restrict__ final_T, int* __restrict__ n_contrib,
                    int* __restrict__ pid, float* __restrict__ pwp, float* __restrict__ pw,
                    float4* __restrict__ zero_conic, int xcd_mode, int cull, uint32_t* __restrict__ lazy_state, int lazy,
                    uint64_t* __restrict__ masks, uint32_t* __restrict__ hdr_w) {
  if (lr_bail(state, capacity)) return;
  if (lazy == 2 && !lazy_state[LR_HDR_OPEN]) return;         // nobody parked (lazy_state: the tile state again, through the pointer these kernels WRITE sorted[] / open[] / the flag with)
  if (masks && blockIdx.x == 0 && threadIdx.x == 0) hdr_w[LR_HDR_MASKS] = LR_MASK_FORM_QUAD;
  const uint32_t tile = lr_tile_of_block(blockIdx.x, tiles, v.gx, v.gy, xcd_mode, state);
  if (tile >= tiles) return;
  const uint32_t* offsets = state + lr_offsets_off(tiles);
  uint32_t beg = offsets[tile];
  uint32_t end = offsets[tile + 1], first;
  bool clamped;
  const int lane = threadIdx.x & 63, quad = threadIdx.x >> 6;
  uint64_t* const mrow_out = masks ? masks + 4 * lr_mask_slot(beg, tile) + quad : nullptr;
  __shared__ uint32_t lr_wmax_q[4][64];                      // per wave and chunk: the running maximum of alpha T of every entry
  uint32_t* const wmx = lr_wmax_q[quad];
  if (EXTRAS) wmx[lane] = 0u;
  uint32_t id_prev = 0xffffffffu;                            // the ids of the chunk whose commit is pending (lr_fwd_commit_chunk)
  if (!lr_lazy_range(lazy_state + lr_sorted_off(tiles), tiles, lazy, tile, quad, beg, end, first, clamped)) return;
  const int tx = tile % (uint32_t)v.gx, ty = tile / (uint32_t)v.gx;
  const int qx0 = tx * 16 + (quad & 1) * 8, qy0 = ty * 16 + (quad >> 1) * 8;
  const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
  const float pxf = (float)px, pyf = (float)py;
  const float bx0 = (float)qx0, bx1 = (float)(qx0 + 7), by0 = (float)qy0, by1 = (float)(qy0 + 7);
  const bool inside = (px < v.W) && (py < v.H);
  const size_t pix = inside ? (size_t)py * v.W + px : 0;
  bool done = !inside;
  float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, wmax = 0.f;
  int wid = -1, last = 0;
  if (lazy == 2) {                                           // second pass: this wave parked at list position `first`
    if (inside) lr_lazy_resume<EXTRAS>(v, pix, done, T, C0, C1, C2, last, wid, wmax, image, final_T, n_contrib, pid, pwp);
    beg += first;
  }

  // Software pipeline over 64-entry chunks: ids are fetched two chunks ahead and records one chunk ahead, so
  // the dependent id -> record gather of chunk c+1 is in flight while chunk c is composited.
  const uint32_t nchunks = (end - beg + 63u) >> 6;
  auto load_id = [&](uint32_t c) -> uint32_t {
    const uint32_t idx = beg + c * 64u + (uint32_t)lane;
    return (c < nchunks && idx < end) ? plist[idx] : 0xffffffffu;
  };
  uint32_t id_n = load_id(0), id_nn = load_id(1);
  float4 g0_n = {0.f, 0.f, 0.f, 0.f};
  float2 g1_n = {0.f, 0.f};  // (conic C, opacity): all the support test needs of q1
  if (id_n != 0xffffffffu) { g0_n = geom[LR_REC_QUADS * (size_t)id_n]; g1_n = *reinterpret_cast<const float2*>(geom + LR_REC_QUADS * (size_t)id_n + 1); }

  for (uint32_t ch = 0; ch < nchunks; ch++) {
    if (__all(done)) break;
    lr_fwd_commit_chunk<EXTRAS>(wmx, lane, id_prev, pw, zero_conic);   // the previous chunk's, in front of this chunk's loads
    const uint32_t id = id_n;
    id_prev = id;
    const float4 g0 = g0_n;
    const float2 g1 = g1_n;
    id_n = id_nn;
    id_nn = load_id(ch + 2);
    if (id_n != 0xffffffffu) { g0_n = geom[LR_REC_QUADS * (size_t)id_n]; g1_n = *reinterpret_cast<const float2*>(geom + LR_REC_QUADS * (size_t)id_n + 1); }
    const bool rel = (id != 0xffffffffu) && (cull ? lr_support_hits(g0, g1, bx0, bx1, by0, by1) : true);
    uint64_t todo = __ballot(rel);
    const int pos0 = (int)(first + ch * 64u);
    if (mrow_out && lane == 0) mrow_out[4 * (size_t)(pos0 >> 6)] = todo;   // for the reverse walk (hit masks, above)
    // Two list entries per iteration, branch-free: the two alpha evaluations (power + exp polynomial, ~20 VALU
    // each) are independent, so one wave can issue them back to back instead of waiting out each dependent
    // result; T / done / last are then applied in list order.  Lane predicates stay in SGPR lane masks
    // (v_cmp -> s_and/s_or -> v_cndmask), no exec-mask juggling.
    while (todo) {
      const int j0 = __builtin_ctzll(todo);
      todo &= todo - 1;
      const bool has1 = todo != 0;
      const int j1 = has1 ? __builtin_ctzll(todo) : j0;
      todo &= todo - 1;  // no-op when todo == 0
      // The two entries' records come straight from memory through the scalar cache (uniform address -> s_load):
      // no VALU broadcast at all.  -0.5 and the sign of B move to the pixel side, where they are exact scalings:
      // fma(A*dx, -0.5*dx, .) rounds the same real number as fma((-0.5*A)*dx, dx, .).
      const int gid0 = lr_readlane_i((int)id, j0), gid1 = lr_readlane_i((int)id, j1);
      const float4* __restrict__ r0 = geom + LR_REC_QUADS * (size_t)(uint32_t)gid0;
      const float4* __restrict__ r1 = geom + LR_REC_QUADS * (size_t)(uint32_t)gid1;
      const float4 q00 = r0[0], q01 = r0[1], q10 = r1[0], q11 = r1[1];
      const float cbl0 = reinterpret_cast<const float*>(r0)[8], cbl1 = reinterpret_cast<const float*>(r1)[8];
      // keep the colour loads in this one batch of scalar loads (left alone the compiler sinks them below the hit test:
      // a second, exposed scalar-cache round trip in every contributing iteration)
      asm volatile("" : : "s"(q01.z), "s"(q01.w), "s"(cbl0), "s"(q11.z), "s"(q11.w), "s"(cbl1));
      const float op0 = q01.y, op1 = q11.y;
      const lr_f2 dx2 = lr_f2{q00.x, q10.x} - pxf, dy2 = lr_f2{q00.y, q10.y} - pyf;
      const lr_f2 hdx2 = dx2 * -0.5f, hdy2 = dy2 * -0.5f;
      const lr_f2 bdx = lr_f2{q00.w, q10.w} * dx2;
      const lr_f2 pw2 = lr_fma2(lr_f2{q00.z, q10.z} * dx2, hdx2,
                                lr_fma2(lr_f2{q01.x, q11.x} * dy2, hdy2, lr_f2{-bdx.x, -bdx.y} * dy2));
      const lr_f2 al2 = lr_f2{op0, op1} * lr_exp2(pw2);
      const float power0 = pw2.x, power1 = pw2.y;
      const float alpha0 = fminf(0.99f, al2.x), alpha1 = fminf(0.99f, al2.y);
      // entry j0
      const bool ok0 = !done & !(power0 > 0.f) & !(alpha0 < 1.0f / 255.0f);
      const float test0 = T * (1.f - alpha0);
      const bool stop0 = ok0 & (test0 < 0.0001f);
      const bool acc0 = ok0 & !stop0;
      const float w0 = acc0 ? alpha0 * T : 0.f;
      T = acc0 ? test0 : T;
      last = acc0 ? pos0 + j0 + 1 : last;
      done = done | stop0;
      // entry j1 (masked out when the chunk had an odd number of relevant entries)
      const bool ok1 = has1 & !done & !(power1 > 0.f) & !(alpha1 < 1.0f / 255.0f);
      const float test1 = T * (1.f - alpha1);
      const bool stop1 = ok1 & (test1 < 0.0001f);
      const bool acc1 = ok1 & !stop1;
      const float w1 = acc1 ? alpha1 * T : 0.f;
      T = acc1 ? test1 : T;
      last = acc1 ? pos0 + j1 + 1 : last;
      done = done | stop1;
      // an accumulating lane has w > 0 (alpha >= 1/255, T > 1e-4): testing the VGPR gives the wave-level predicate in one
      // v_cmp, where a ballot of the lane-mask predicate costs a v_cndmask + v_cmp
      const bool hit0 = __builtin_amdgcn_ballot_w64(w0 > 0.f) != 0, hit1 = __builtin_amdgcn_ballot_w64(w1 > 0.f) != 0;
      if (hit0) {
        const float cr = q01.z, cg = q01.w, cbl = cbl0;
        const int gid = gid0;
        if (acc0) { C0 = lr_fma(cr, w0, C0); C1 = lr_fma(cg, w0, C1); C2 = lr_fma(cbl, w0, C2); }
        if (EXTRAS) {
          if (w0 > wmax) { wmax = w0; wid = gid; }
          const uint32_t m = lr_wave_umax_to63(__float_as_uint(w0));  // w >= 0: unsigned order == float order
          if (lane == 63) atomicMax(&wmx[j0], m);                     // (LDS; point_weight and the row clear: lr_fwd_commit_chunk)
        }
      }
      if (hit1) {
        const float cr = q11.z, cg = q11.w, cbl = cbl1;
        const int gid = gid1;
        if (acc1) { C0 = lr_fma(cr, w1, C0); C1 = lr_fma(cg, w1, C1); C2 = lr_fma(cbl, w1, C2); }
        if (EXTRAS) {
          if (w1 > wmax) { wmax = w1; wid = gid; }
          const uint32_t m = lr_wave_umax_to63(__float_as_uint(w1));
          if (lane == 63) atomicMax(&wmx[j1], m);
        }
      }
      if (!(hit0 | hit1) && __all(done)) break;
    }
  }
  lr_fwd_commit_chunk<EXTRAS>(wmx, lane, id_prev, pw, zero_conic);   // the last chunk's
  if (clamped && !__all(done)) {                             // out of ordered entries with a pixel open: to be continued
    if (lane == 0) { atomicOr(lazy_state + lr_sorted_off(tiles) + tiles + tile, 1u << quad); atomicOr(lazy_state + LR_HDR_OPEN, 1u); }   // open[tile], header
    if (inside) lr_lazy_park<EXTRAS>(v, pix, done, T, C0, C1, C2, last, wid, wmax, image, final_T, n_contrib, pid, pwp);
    return;
  }

  if (inside) {
    const size_t plane = (size_t)v.W * v.H;
    final_T[pix] = T;
    n_contrib[pix] = last;
    image[pix] = lr_fma(T, v.bg[0], C0);
    image[plane + pix] = lr_fma(T, v.bg[1], C1);
    image[2 * plane + pix] = lr_fma(T, v.bg[2], C2);
    if (EXTRAS) { pid[pix] = wid; pwp[pix] = wmax; }
  }
}

// ---- backward ---------------------------------------------------------------------------------------------
// Packed wave64 reduction of 9 per-lane sums.  v_permlane32_swap(a,b) leaves a=[a.lo,b.lo], b=[a.hi,b.hi], so
// ONE swap + ONE add halves two values at once (a's sum in lanes 0-31, b's in 32-63); v_permlane16_swap does
// the same between odd/even 16-lane rows; the last four levels are DPP row rotations.  Result:
//   r0 rows 0..3 = sums of (v0, v2, v1, v3);  r1 rows 0..3 = sums of (v4, v6, v5, v7);  r2 = sum of v8
// (every lane of a row holds that row's total).
LR_DEV float lr_swap_add32(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
LR_DEV float lr_swap_add16(float a, float b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
LR_DEV float lr_row_ror_add(float v) {
  int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
  return v + __int_as_float(moved);
}
LR_DEV float lr_row_sum(float v) {
  v = lr_row_ror_add<0x128>(v);  // row_ror:8
  v = lr_row_ror_add<0x124>(v);  // row_ror:4
  v = lr_row_ror_add<0x122>(v);  // row_ror:2
  v = lr_row_ror_add<0x121>(v);  // row_ror:1
  return v;
}
LR_DEV void lr_reduce9(const float v[9], float& r0, float& r1, float& r2) {
  const float p0 = lr_swap_add32(v[0], v[1]), p1 = lr_swap_add32(v[2], v[3]);
  const float p2 = lr_swap_add32(v[4], v[5]), p3 = lr_swap_add32(v[6], v[7]);
  const float s8 = lr_swap_add32(v[8], v[8]);
  r0 = lr_row_sum(lr_swap_add16(p0, p1));
  r1 = lr_row_sum(lr_swap_add16(p2, p3));
  r2 = lr_row_sum(lr_swap_add16(s8, s8));
}

template <bool MASKS>
__global__ void __launch_bounds__(256) LR_OCC_BWD
lr_blend_bwd_kernel(LrView v, const float4* __restrict__ geom, const uint32_t* __restrict__ state,
                    uint32_t tiles, const uint32_t* __restrict__ plist, uint32_t capacity,
                    const float* __restrict__ final_T, const int* __restrict__ n_contrib,
                    const float* __restrict__ dL_dimage, float* __restrict__ acc_rows,
                    int xcd_mode, int cull, const uint64_t* __restrict__ masks) {
  if (lr_bail(state, capacity)) return;
  const uint32_t tile = lr_tile_of_block(blockIdx.x, tiles, v.gx, v.gy, xcd_mode, state);
  if (tile >= tiles) return;
  const uint32_t* offsets = state + lr_offsets_off(tiles);
  const uint32_t beg = offsets[tile];
  const int lane = threadIdx.x & 63, quad = threadIdx.x >> 6;
  constexpr bool use_masks = MASKS;   // the forward left its quadrant ballots in `masks` (the host checked the form: lr_launch_blend_bwd)
  const int tx = tile % (uint32_t)v.gx, ty = tile / (uint32_t)v.gx;
  const int qx0 = tx * 16 + (quad & 1) * 8, qy0 = ty * 16 + (quad >> 1) * 8;
  const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
  const float pxf = (float)px, pyf = (float)py;
  const float bx0 = (float)qx0, bx1 = (float)(qx0 + 7), by0 = (float)qy0, by1 = (float)(qy0 + 7);
  const bool in = (px < v.W) && (py < v.H);
  const size_t plane = (size_t)v.W * v.H;
  const size_t pix = in ? (size_t)py * v.W + px : 0;
  const float sx = 0.5f * (float)v.W, sy = 0.5f * (float)v.H;
  const float Tf = in ? final_T[pix] : 0.f;
  const int lastc = in ? n_contrib[pix] : 0;
  const float dp0 = in ? dL_dimage[pix] : 0.f;
  const float dp1 = in ? dL_dimage[plane + pix] : 0.f;
  const float dp2 = in ? dL_dimage[2 * plane + pix] : 0.f;
  const float bgdot = lr_fma(v.bg[0], dp0, lr_fma(v.bg[1], dp1, v.bg[2] * dp2));
  float T = Tf, acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;  // acc = colour composited behind the current entry
  int maxc = lastc;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) maxc = max(maxc, __shfl_xor(maxc, off));
  maxc = lr_readlane_i(maxc, 0);  // wave-uniform: deepest contributor among this quadrant's pixels

  // Destinations of the packed reduction (see lr_reduce9): the Gaussian's 64-byte accumulator row (slots 0-1 mean x y,
  // 2-4 conic A B C, 5 opacity, 6-8 colour r g b), as loop-invariant per-lane slots so that the per-Gaussian address is one
  // multiply-add, no divergence.  Both instructions land in ONE line: the memory side charges an atomic per 64-byte line,
  // not per lane (tools/micro/atomic_lines.hip).
  //   atomic #1, lanes 0/16/32/48 (rows 0..3): colour r,g,b (slots 6-8) and opacity (slot 5)
  //   atomic #2, rows 0..3: mean x,y (slots 0-1), conic A,B (slots 2-3); lane 1 additionally carries conic C (slot 4)
  const int row = lane >> 4;
  float* const base0 = acc_rows + ((row < 3) ? 6 + row : 5);
  float* const base1 = acc_rows + ((lane == 1) ? 4 : row);
  const bool lead = (lane & 15) == 0;

  // Reverse walk in 64-entry chunks from the deepest contributor; lane l of chunk ch holds list position
  // top-1 - 64*ch - l, top = maxc -- or, with the forward's hit masks, maxc rounded up to the forward's chunk grid: chunk ch
  // is then the forward's chunk top/64 - 1 - ch with its lanes reversed, and its visits are the set bits of the forward's
  // (bit-reversed) ballot in front of maxc: no record is gathered per lane, no support test runs.
  const int top = use_masks ? ((maxc + 63) & ~63) : maxc;
  const uint32_t nchunks = ((uint32_t)top + 63u) >> 6;
  auto load_id = [&](uint32_t c) -> uint32_t {
    const int pos = top - 1 - (int)(c * 64u) - lane;
    return (c < nchunks && pos >= 0 && pos < maxc) ? plist[beg + (uint32_t)pos] : 0xffffffffu;
  };
  const uint64_t* const mbase = use_masks ? masks + 4 * lr_mask_slot(beg, tile) + __builtin_amdgcn_readfirstlane(quad) : nullptr;
  auto load_mask = [&](uint32_t c) -> uint64_t {             // (uniform address: a scalar load)
    return (use_masks && c < nchunks) ? mbase[4 * (size_t)(((uint32_t)top >> 6) - 1u - c)] : 0ull;
  };
  uint32_t id_n = load_id(0), id_nn = load_id(1);
  uint64_t mk_n = load_mask(0), mk_nn = load_mask(1);
  float4 g0_n = {0.f, 0.f, 0.f, 0.f};
  float2 g1_n = {0.f, 0.f};  // (conic C, opacity): all the support test needs of q1
  if (!use_masks && id_n != 0xffffffffu) { g0_n = geom[LR_REC_QUADS * (size_t)id_n]; g1_n = *reinterpret_cast<const float2*>(geom + LR_REC_QUADS * (size_t)id_n + 1); }

  for (uint32_t ch = 0; ch < nchunks; ch++) {
    const int hi = top - (int)(ch * 64u);
    const uint32_t id = id_n;
    const float4 g0 = g0_n;
    const float2 g1 = g1_n;
    const uint64_t mk = mk_n;
    id_n = id_nn;
    id_nn = load_id(ch + 2);
    mk_n = mk_nn;
    mk_nn = load_mask(ch + 2);
    uint64_t todo;
    if (use_masks) {
      todo = lr_mask_before(__builtin_bitreverse64(mk), hi, maxc);
    } else {
      if (id_n != 0xffffffffu) { g0_n = geom[LR_REC_QUADS * (size_t)id_n]; g1_n = *reinterpret_cast<const float2*>(geom + LR_REC_QUADS * (size_t)id_n + 1); }
      const bool rel = (id != 0xffffffffu) && (cull ? lr_support_hits(g0, g1, bx0, bx1, by0, by1) : true);
      todo = __ballot(rel);
    }
    // Two entries per iteration: both alpha evaluations are issued together (independent chains), the
    // gradient bodies then run in list order.
    while (todo) {
      const int j0 = __builtin_ctzll(todo);
      todo &= todo - 1;
      const bool has1 = todo != 0;
      const int j1 = has1 ? __builtin_ctzll(todo) : j0;
      todo &= todo - 1;
      const int gid0 = lr_readlane_i((int)id, j0), gid1 = lr_readlane_i((int)id, j1);
      const float4* __restrict__ rr0 = geom + LR_REC_QUADS * (size_t)(uint32_t)gid0;
      const float4* __restrict__ rr1 = geom + LR_REC_QUADS * (size_t)(uint32_t)gid1;
      const float4 q00 = rr0[0], q01 = rr0[1], q10 = rr1[0], q11 = rr1[1];
      const float cbl0 = reinterpret_cast<const float*>(rr0)[8], cbl1 = reinterpret_cast<const float*>(rr1)[8];
      // keep the colour loads in this first batch of scalar loads: left alone the compiler sinks them below the hit
      // test, i.e. a second, fully exposed scalar-cache round trip in every contributing iteration
      asm volatile("" : : "s"(q01.z), "s"(q01.w), "s"(cbl0), "s"(q11.z), "s"(q11.w), "s"(cbl1));
      const float op0 = q01.y, op1 = q11.y;
      const lr_f2 dx2 = lr_f2{q00.x, q10.x} - pxf, dy2 = lr_f2{q00.y, q10.y} - pyf;
      const lr_f2 hdx2 = dx2 * -0.5f, hdy2 = dy2 * -0.5f;
      const lr_f2 bdx = lr_f2{q00.w, q10.w} * dx2;
      const lr_f2 pw2 = lr_fma2(lr_f2{q00.z, q10.z} * dx2, hdx2,
                                lr_fma2(lr_f2{q01.x, q11.x} * dy2, hdy2, lr_f2{-bdx.x, -bdx.y} * dy2));
      const lr_f2 G2 = lr_exp2(pw2);
      const lr_f2 al2 = lr_f2{op0, op1} * G2;
      const float power0 = pw2.x, power1 = pw2.y, G0 = G2.x, G1 = G2.y;
      const float alpha0 = fminf(0.99f, al2.x), alpha1 = fminf(0.99f, al2.y);
      const int k0 = hi - 1 - j0, k1 = hi - 1 - j1;  // 0-based positions in the tile list
      const bool hit0 = (k0 < lastc) & !(power0 > 0.f) & !(alpha0 < 1.0f / 255.0f);
      const bool hit1 = has1 & (k1 < lastc) & !(power1 > 0.f) & !(alpha1 < 1.0f / 255.0f);
      const lr_f2 alpha = {hit0 ? alpha0 : 0.f, hit1 ? alpha1 : 0.f};
      // (a contributing lane has alpha >= 1/255: testing the VGPR is one v_cmp, a ballot of the lane-mask predicate two ops)
      const bool any0 = __builtin_amdgcn_ballot_w64(alpha.x > 0.f) != 0, any1 = __builtin_amdgcn_ballot_w64(alpha.y > 0.f) != 0;
      if (!(any0 | any1)) continue;
      // Both entries of the pair go through ONE 2-wide body (v_pk_* f32: two entries per instruction).  Lanes that do
      // not contribute run it with alpha = G = 0, which leaves their state untouched exactly (T*1, 0*c + 1*acc) and
      // makes all nine of their partial sums exact zeros -- no exec masking, no zero-initialised accumulators.  Only T
      // and the colour behind the current entry (acc <- alpha c + (1-alpha) acc) chain from entry 0 to entry 1.
      const lr_f2 G = {hit0 ? G0 : 0.f, hit1 ? G1 : 0.f};
      const lr_f2 om = 1.f - alpha;
      lr_f2 rc = {__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
      rc = lr_fma2(lr_fma2(-om, rc, lr_f2{1.f, 1.f}), rc, rc);  // one Newton step on v_rcp_f32
      const float Ta = T * rc.x, Tb = Ta * rc.y;                // T in front of entry 0 / entry 1
      const lr_f2 T2 = {Ta, Tb};
      T = Tb;
      const lr_f2 w = alpha * T2;
      const lr_f2 cr = {q01.z, q11.z}, cg = {q01.w, q11.w}, cbl = {cbl0, cbl1};
      const float a0r = lr_fma(alpha.x, cr.x, om.x * acc0), a0g = lr_fma(alpha.x, cg.x, om.x * acc1),
                  a0b = lr_fma(alpha.x, cbl.x, om.x * acc2);   // colour behind entry 1
      lr_f2 dL_dalpha = lr_fma2(cr - lr_f2{acc0, a0r}, lr_f2{dp0, dp0},
                                lr_fma2(cg - lr_f2{acc1, a0g}, lr_f2{dp1, dp1}, (cbl - lr_f2{acc2, a0b}) * dp2));
      dL_dalpha = lr_fma2(dL_dalpha, T2, -(Tf * rc) * bgdot);
      acc0 = lr_fma(alpha.y, cr.y, om.y * a0r);
      acc1 = lr_fma(alpha.y, cg.y, om.y * a0g);
      acc2 = lr_fma(alpha.y, cbl.y, om.y * a0b);
      const lr_f2 Ar = {q00.z, q10.z}, Br = {q00.w, q10.w}, Cr = {q01.x, q11.x};
      const lr_f2 dL_dG = lr_f2{op0, op1} * dL_dalpha;
      const lr_f2 gdx = G * dx2, gdy = G * dy2;
      const lr_f2 dG_ddx = lr_fma2(-Ar, gdx, -(Br * gdy));     // -gdx*A - gdy*B
      const lr_f2 dG_ddy = lr_fma2(-Cr, gdy, -(Br * gdx));     // -gdy*C - gdx*B
      // slot order chosen so that lr_reduce9's rows land on contiguous destinations:
      //   r0 rows = (s0,s2,s1,s3) = (col r, col g, col b, opacity); r1 rows = (s4,s6,s5,s7) = (mean x, mean y, conic A, conic B); r2 = conic C
      lr_f2 s2[9];
      s2[0] = w * dp0; s2[2] = w * dp1; s2[1] = w * dp2; s2[3] = G * dL_dalpha;
      s2[4] = dL_dG * dG_ddx * sx; s2[6] = dL_dG * dG_ddy * sy;
      s2[5] = -0.5f * gdx * dx2 * dL_dG; s2[7] = -gdx * dy2 * dL_dG;
      s2[8] = -0.5f * gdy * dy2 * dL_dG;
#pragma unroll
      for (int e = 0; e < 2; e++) {
        if (!(e ? any1 : any0)) continue;                       // wave-uniform: nothing to commit for this entry
        const int gid = e ? gid1 : gid0;
        float sv[9];
#pragma unroll
        for (int m = 0; m < 9; m++) sv[m] = e ? s2[m].y : s2[m].x;
        float r0, r1, r2;
        lr_reduce9(sv, r0, r1, r2);
        if (lead) atomicAdd(base0 + (size_t)gid * LOGRAST_BWD_ROW_FLOATS, r0);
        if (lead || lane == 1) atomicAdd(base1 + (size_t)gid * LOGRAST_BWD_ROW_FLOATS, lane == 1 ? r2 : r1);
      }
    }
  }
}

// ---- backward, row-split form -----------------------------------------------------------------------------------
// Same workgroup / wave -> quadrant mapping as above, but the wave's four 16-lane ROWS (the unit DPP operates on) each
// own a 4x4 pixel block of the quadrant and walk the chunk's entries on their own: per iteration every row takes its own
// next two relevant entries, so one pass of the 2-wide body serves up to eight (Gaussian, block) pairs instead of two
// (Gaussian, quadrant) pairs.  What that buys on tiny splats (support radius ~4 px against the 8 px quadrant):
//   * a visit evaluates 16 pixels of which ~8 contribute, instead of 64 of which ~18 do (tools/visit_stats.py: 0.21
//     row-iterations per list entry against 0.34 quadrant visits);
//   * the nine per-Gaussian sums are reduced over 16 lanes with DPP row operations only -- row_mirror, row_half_mirror and
//     two quad permutes, packed two values per register through the 16 -> 8 and 8 -> 4 levels with bank-masked moves:
//     ~50 VALU for BOTH entries of all four rows against 2 x 28 for one entry each of the whole wave.
// The entries of a chunk are staged once per wave in LDS (48 B each; slot 64 is an all-zero entry that rows without work
// read: alpha = 0 contributes nothing, branch-free); a row's lanes read their entry with three broadcast ds_read_b128.
// Committing the sums is what decides this form.  A row visit is its own commit (2.5x as many as quadrant visits), and the
// memory side charges an atomic per 64-byte LINE it touches (tools/micro/atomic_lines.hip: 17-21 G lines/s chip-wide
// whether 1, 9 or 16 lanes share the line).  With the sums in four separate arrays a row visit was 5 line operations:
// C2 978 us (220 us with the atomics removed).  Through LDS instead: ds_add_f32 into a per-wave table runs at ~1 lane per
// clock per CU (344 us); per-row cells written with plain stores and summed once per chunk: 322 us (274 without the
// commits) -- the table's traffic eats what the rows gain.  With ONE 64-byte accumulator row per Gaussian (the layout
// lograst_backward now uses) a row visit is ONE line operation: the nine sums of an entry are moved into nine lanes of
// the row (two v_cndmask) and leave in one atomic instruction per entry.
#define LR_RB_SLOT 3      // float4 per staged entry: (mx, my, A, B) (C, opacity, r, g) (b, id, -, -)
template <int CTRL>
LR_DEV float lr_dpp_perm(float x) {   // full-mask permutation within the row (all the controls used are self-inverse)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int BANKS>
LR_DEV float lr_bank_select(float keep, float take) {   // lanes of the 4-lane banks in BANKS <- take, the others keep
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(take), 0xE4, 0xf, BANKS, false));
}
// lanes 0-7 of the row: the eight pair sums of a; lanes 8-15: those of b (row_mirror: lane i <-> 15 - i)
LR_DEV float lr_row_pair8(float a, float b) {
  return lr_bank_select<0xc>(a + lr_dpp_perm<0x140>(a), b + lr_dpp_perm<0x140>(b));
}
// banks 0 and 2 keep p's four-lane sums, banks 1 and 3 take q's (row_half_mirror: i <-> 7 - i inside each half)
LR_DEV float lr_row_pair4(float p, float q) {
  return lr_bank_select<0xa>(p + lr_dpp_perm<0x141>(p), q + lr_dpp_perm<0x141>(q));
}
LR_DEV float lr_quad_total(float x) {   // every lane of a quad <- the quad's total
  x = x + lr_dpp_perm<0xB1>(x);         // quad_perm [1,0,3,2]
  return x + lr_dpp_perm<0x4E>(x);      // quad_perm [2,3,0,1]
}

// The commit of a row visit, as the bare instruction.  Written with atomicAdd the pass loop "contains a VMEM access that may
// load" (LLVM marks every atomic so), and SIInsertWaitcnts then flushes vmcnt in the loop's preheader (shouldFlushVmCnt)
// -- i.e. it waits for the records and ids of the NEXT chunk, requested a few dozen instructions earlier, before the first
// pass of THIS chunk: one exposed memory round trip per chunk and wave, the prefetch pipeline never overlaps anything.
// Inline asm is invisible to that pass.  Safe: the instruction returns nothing; vmcnt counts it like any store, and an
// uncounted operation can only make the compiler's own `s_waitcnt vmcnt(n)` wait longer than needed, never shorter (the
// counter retires in order: at most n outstanding means everything but the youngest n has completed).
LR_DEV void lr_atomic_add_noret(float* p, float x) {
  asm volatile("global_atomic_add_f32 %0, %1, off" : : "v"(p), "v"(x) : "memory");
}

// The next entry of a row's hit mask, per lane (every lane of a 16-lane row holds the row's mask): position of the lowest
// set bit, 64 = none (the all-zero staging slot); the bit is cleared.  Vector instructions (two v_ffbl, a few selects, a
// 64-bit add / and): the scalar form -- four masks walked by s_ff1 / s_and / s_cselect chains, ~85 scalar instructions in
// front of every pass -- was as long as the pass's vector work and strictly serial (round 4: SQ_INSTS_SALU 1.9e8 against
// SQ_INSTS_VALU 2.1e8 per launch of the row-split forward).
LR_DEV uint32_t lr_take_bit(uint64_t& m) {
  const uint32_t j = min((uint32_t)(__ffsll((long long)m) - 1), 64u);   // __ffsll(0) = 0 -> 0xffffffff -> 64
  m &= m - 1ull;                                                        // (0 stays 0)
  return j;
}
LR_DEV uint64_t lr_row_mask(int row, uint64_t m0, uint64_t m1, uint64_t m2, uint64_t m3) {
  return row == 0 ? m0 : (row == 1 ? m1 : (row == 2 ? m2 : m3));
}

template <bool MASKS>
__global__ void __launch_bounds__(256) LR_OCC_BWD_ROWS
lr_blend_bwd_rows_kernel(LrView v, const float4* __restrict__ geom, const uint32_t* __restrict__ state,
                         uint32_t tiles, const uint32_t* __restrict__ plist, uint32_t capacity,
                         const float* __restrict__ final_T, const int* __restrict__ n_contrib,
                         const float* __restrict__ dL_dimage, float* __restrict__ acc_rows,
                         int xcd_mode, int cull, int block_test, const uint64_t* __restrict__ masks LR_ABLATE_PARAM) {
  __shared__ float4 lr_stage[4][65 * LR_RB_SLOT];
  if (lr_bail(state, capacity)) return;
  const uint32_t tile = lr_tile_of_block(blockIdx.x, tiles, v.gx, v.gy, xcd_mode, state);
  if (tile >= tiles) return;
  const uint32_t* offsets = state + lr_offsets_off(tiles);
  const uint32_t beg = offsets[tile];
  const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
  // the forward's block masks of this wave's chunks (hit masks, above), if it left any in this form
  constexpr bool use_masks = MASKS;   // the forward left its block ballots in `masks` (the host checked the form: lr_launch_blend_bwd)
  const int row = lane >> 4, li = lane & 15;
  const int tx = tile % (uint32_t)v.gx, ty = tile / (uint32_t)v.gx;
  const int qx0 = tx * 16 + (wq & 1) * 8, qy0 = ty * 16 + (wq >> 1) * 8;
  const int px = qx0 + (row & 1) * 4 + (li & 3), py = qy0 + (row >> 1) * 4 + (li >> 2);
  const float pxf = (float)px, pyf = (float)py;
  const bool in = (px < v.W) && (py < v.H);
  const size_t plane = (size_t)v.W * v.H;
  const size_t pix = in ? (size_t)py * v.W + px : 0;
  const float sx = 0.5f * (float)v.W, sy = 0.5f * (float)v.H;
  const float Tf = in ? final_T[pix] : 0.f;
  const int lastc = in ? n_contrib[pix] : 0;
  const float dp0 = in ? dL_dimage[pix] : 0.f;
  const float dp1 = in ? dL_dimage[plane + pix] : 0.f;
  const float dp2 = in ? dL_dimage[2 * plane + pix] : 0.f;
  const float bgdot = lr_fma(v.bg[0], dp0, lr_fma(v.bg[1], dp1, v.bg[2] * dp2));
  float T = Tf, acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
  // deepest contributor of every row's block (entries behind it are not that row's business) and of the wave
  int rmax = lastc;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) rmax = max(rmax, __shfl_xor(rmax, off));
  const int rm0 = lr_readlane_i(rmax, 0), rm1 = lr_readlane_i(rmax, 16), rm2 = lr_readlane_i(rmax, 32),
            rm3 = lr_readlane_i(rmax, 48);
  const int maxc = max(max(rm0, rm1), max(rm2, rm3));
  if (maxc == 0) return;

  // The chunk's entries staged FIELD-major (round 6): sf[f * 65 + j] = field f of entry j (0 mx, 1 my, 2 A, 3 B, 4 C, 5 opacity,
  // 6-8 colour, 9 id bits); slot 64 = the all-zero entry of rows without work.  The pass loop packs the SAME field of its two
  // entries into one 64-bit register pair (v_pk_*: entry a in the low half, entry b in the high half): with 48-byte entries read
  // as three ds_read_b128 each, every such pair cost a v_mov per half (27 of the loop's 204 VALU instructions were plain
  // moves); a ds_read_b32 per field lands in the half it is used in.
  float* const sf = reinterpret_cast<float*>(lr_stage[wq]);
#define LR_SF(f, j) sf[(f) * 65 + (j)]
  const uint32_t sf_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)sf;   // its LDS byte address
  if (lane < 10) LR_SF(lane, 64) = lane == 9 ? __uint_as_float(0xffffffffu) : 0.f;   // the all-zero entry

  // Where a row's sums go: after the packed reductions (see the loop) every lane of a quad holds the quad's total of
  //   S1: quads (col r, col b, col g, opacity)   S2: quads (mean x, conic A, mean y, conic B)   S3: lanes 0-7 / 8-15: conic C of entry 0 / 1
  // Lanes (lane & 3) == 0 take S1, == 1 take S2, lane 2 (entry 0) / lane 10 (entry 1) take S3: nine lanes of the row
  // carry the entry's nine sums into the Gaussian's accumulator row in one instruction (slots: 0-1 mean x y, 2-4 conic
  // A B C, 5 opacity, 6-8 colour r g b).
  const int qd = li >> 2, sel = li & 3;
  const int slot = sel == 0 ? (qd == 0 ? 6 : (qd == 1 ? 8 : (qd == 2 ? 7 : 5)))
                            : (sel == 1 ? (qd == 0 ? 0 : (qd == 1 ? 2 : (qd == 2 ? 1 : 3))) : 4);
  const bool on_a = sel < 2 || li == 2, on_b = sel < 2 || li == 10;
  float* const dst = acc_rows + slot;

  // the four blocks of this quadrant (wave-uniform), for the support tests
  const float bx[2] = {(float)qx0, (float)(qx0 + 4)}, by[2] = {(float)qy0, (float)(qy0 + 4)};

  // With the forward's masks the walk runs on the forward's chunk grid: top = maxc rounded up to a multiple of 64, chunk ch
  // = the forward's chunk top/64 - 1 - ch with its lanes reversed (lane l: list position top-1 - 64 ch - l), a row's visits =
  // the set bits of its (bit-reversed) block mask in front of the row's deepest contributor; only entries with a bit in
  // one of the four masks are gathered and staged.  Without masks: top = maxc and the tests run here, as before.
  const int top = use_masks ? ((maxc + 63) & ~63) : maxc;
  const uint32_t nchunks = ((uint32_t)top + 63u) >> 6;
  // The loads of this pipeline are UNCONDITIONAL (lanes without an entry read a harmless address and the result is replaced
  // by a select): as exec-masked branches every load had a `keep the old value` copy merged in behind it, and the waitcnt
  // insertion -- which joins the loop's entry state with its steady state -- then put s_waitcnt vmcnt(0) right behind the id
  // load at the top of every chunk: one exposed memory round trip per chunk and wave (rounds 3-5 shipped that).  Order per
  // chunk: the gather of chunk ch + 1 first (its ids arrived a chunk ago), then the ids and masks of chunk ch + 2.
  auto load_id = [&](uint32_t c) -> uint32_t {
    const int pos = top - 1 - (int)(c * 64u) - lane;
    const bool ok = c < nchunks && pos >= 0 && pos < maxc;
    const uint32_t got = plist[beg + (ok ? (uint32_t)pos : 0u)];
    return ok ? got : 0xffffffffu;
  };
  const uint4* const mbase = use_masks ? reinterpret_cast<const uint4*>(masks + 16 * lr_mask_slot(beg, tile) +
                                                                        4 * __builtin_amdgcn_readfirstlane(wq)) : nullptr;
  struct Masks4 { uint64_t m0, m1, m2, m3; };                // as the forward wrote them: bit j = list position 64 c + j
  auto load_masks = [&](uint32_t c) -> Masks4 {              // (uniform address: scalar loads, 32 bytes per wave and chunk)
    if (!use_masks) return Masks4{0ull, 0ull, 0ull, 0ull};
    const uint32_t cf = ((uint32_t)top >> 6) - 1u - min(c, nchunks - 1u);    // (past the end: the last chunk's once more)
    const uint4* mp = mbase + 8 * (size_t)cf;                // (16 words = 8 uint4 per slot)
    const uint4 lo = mp[0], hi4 = mp[1];
    return Masks4{((uint64_t)lo.y << 32) | lo.x, ((uint64_t)lo.w << 32) | lo.z,
                  ((uint64_t)hi4.y << 32) | hi4.x, ((uint64_t)hi4.w << 32) | hi4.z};
  };
  auto wanted = [&](const Masks4& m) -> bool {               // does any row of this wave visit lane's entry?  (lane l of a
    return !use_masks || (((m.m0 | m.m1 | m.m2 | m.m3) >> (63 - lane)) & 1ull) != 0ull;   // reverse chunk = the forward's bit 63 - l)
  };
  auto gather = [&](uint32_t idv, bool want, float4& q0, float4& q1, float& qc) {
    const float4* rp = geom + LR_REC_QUADS * (size_t)(want ? idv : 0u);       // (record 0: inside the buffer, never used)
    q0 = rp[0]; q1 = rp[1]; qc = reinterpret_cast<const float*>(rp)[8];
  };
  uint32_t id_n = load_id(0), id_nn = load_id(1);
  Masks4 mk_n = load_masks(0), mk_nn = load_masks(1);
  float4 g0_n, g1_n;
  float cb_n;
  gather(id_n, id_n != 0xffffffffu && wanted(mk_n), g0_n, g1_n, cb_n);
  // Everything requested so far has landed before the loop is entered: the waitcnt insertion joins the loop's entry state
  // with its steady state, and with the first gather still in flight at the entry it guards every later write of those
  // registers -- temporaries of the pass loop -- with s_waitcnt vmcnt(4..5), which (the counter retires in order, and the
  // commits of the passes count too) makes the second and third pass of every chunk wait for the NEXT chunk's records.
  LR_LANDED(g0_n, g1_n, cb_n, id_nn);

  for (uint32_t ch = 0; ch < nchunks; ch++) {
    const int hi = top - (int)(ch * 64u);
    const uint32_t id = id_n;
    const float4 g0 = g0_n, g1 = g1_n;
    const float cb = cb_n;
    const Masks4 mk = mk_n;
    id_n = id_nn;
    mk_n = mk_nn;
    gather(id_n, id_n != 0xffffffffu && wanted(mk_n), g0_n, g1_n, cb_n);
    id_nn = load_id(ch + 2);
    mk_nn = load_masks(ch + 2);
    // this chunk's entries -> LDS (the wave's own slots; the previous chunk's reads have all returned)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (id != 0xffffffffu && wanted(mk)) {
      LR_SF(0, lane) = g0.x; LR_SF(1, lane) = g0.y; LR_SF(2, lane) = g0.z; LR_SF(3, lane) = g0.w;
      LR_SF(4, lane) = g1.x; LR_SF(5, lane) = g1.y; LR_SF(6, lane) = g1.z; LR_SF(7, lane) = g1.w;
      LR_SF(8, lane) = cb; LR_SF(9, lane) = __uint_as_float(id);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // which rows does lane's entry concern?  valid, in front of the row's deepest contributor, and able to reach the
    // alpha floor somewhere in the row's 4x4 block (conservative test shared with the binning stage)
    const bool valid = id != 0xffffffffu;
    const int pos = hi - 1 - lane;
    bool r0 = valid & (pos < rm0), r1 = valid & (pos < rm1), r2 = valid & (pos < rm2), r3 = valid & (pos < rm3);
    if (cull && !use_masks) {
      const LrSupport sp = lr_support_prepare(g0.x, g0.y, g0.z, g0.w, g1.x, g1.y);
      if (block_test) {   // the exact ellipse-vs-box test for each of the four blocks (~75 VALU each)
        bool k0, k1, k2, k3;   // (two blocks at a time: lr_support_box2, the decisions of four lr_support_box calls)
        const lr_f2 X0 = {bx[0], bx[1]}, X1 = {bx[0] + 3.f, bx[1] + 3.f};
        lr_support_box2(sp, X0, X1, lr_f2{by[0], by[0]}, lr_f2{by[0] + 3.f, by[0] + 3.f}, k0, k1);
        lr_support_box2(sp, X0, X1, lr_f2{by[1], by[1]}, lr_f2{by[1] + 3.f, by[1] + 3.f}, k2, k3);
        r0 = r0 && k0; r1 = r1 && k1; r2 = r2 && k2; r3 = r3 && k3;
      } else {            // exact test for the quadrant, the support's bounding box against each block (4 compares each)
        const bool q = lr_support_box(sp, bx[0], bx[0] + 7.f, by[0], by[0] + 7.f);
        const bool bb = sp.mode == 2;
        const bool x_lo = !bb || (sp.mx - sp.ex <= bx[0] + 3.f), x_hi = !bb || (sp.mx + sp.ex >= bx[1]);
        const bool y_lo = !bb || (sp.my - sp.ey <= by[0] + 3.f), y_hi = !bb || (sp.my + sp.ey >= by[1]);
        r0 = r0 && q && x_lo && y_lo;
        r1 = r1 && q && x_hi && y_lo;
        r2 = r2 && q && x_lo && y_hi;
        r3 = r3 && q && x_hi && y_hi;
      }
    }
    uint64_t mrow;                                            // this lane's row's hit mask
    if (use_masks) mrow = lr_mask_before(lr_row_mask(row, __builtin_bitreverse64(mk.m0), __builtin_bitreverse64(mk.m1),
                                                     __builtin_bitreverse64(mk.m2), __builtin_bitreverse64(mk.m3)), hi, rmax);   // (rmax: this lane's row's deepest contributor)
    else mrow = lr_row_mask(row, __ballot(r0), __ballot(r1), __ballot(r2), __ballot(r3));
    if (LR_ABLATED(4)) mrow = 0ull;   // experiment builds: the chunk prologue alone
    while (__builtin_amdgcn_ballot_w64(mrow != 0ull) != 0) {
      // every row's next two entries (64 = none: the all-zero slot)
      const uint32_t ja = lr_take_bit(mrow), jb = lr_take_bit(mrow);
      // twenty single-dword LDS reads, each into the register half it is used in (written with plain loads the compiler
      // merges two FIELDS of one entry into a ds_read2_b32 -- pairs within an entry again, and the moves are back)
      float fa[10], fb[10];
      {
        const uint32_t aa = sf_lds + 4u * ja, ab = sf_lds + 4u * jb;
        asm volatile(
            "ds_read_b32 %0, %20\n ds_read_b32 %10, %21\n"
            "ds_read_b32 %1, %20 offset:260\n ds_read_b32 %11, %21 offset:260\n"
            "ds_read_b32 %2, %20 offset:520\n ds_read_b32 %12, %21 offset:520\n"
            "ds_read_b32 %3, %20 offset:780\n ds_read_b32 %13, %21 offset:780\n"
            "ds_read_b32 %4, %20 offset:1040\n ds_read_b32 %14, %21 offset:1040\n"
            "ds_read_b32 %5, %20 offset:1300\n ds_read_b32 %15, %21 offset:1300\n"
            "ds_read_b32 %6, %20 offset:1560\n ds_read_b32 %16, %21 offset:1560\n"
            "ds_read_b32 %7, %20 offset:1820\n ds_read_b32 %17, %21 offset:1820\n"
            "ds_read_b32 %8, %20 offset:2080\n ds_read_b32 %18, %21 offset:2080\n"
            "ds_read_b32 %9, %20 offset:2340\n ds_read_b32 %19, %21 offset:2340\n"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(fa[0]), "=&v"(fa[1]), "=&v"(fa[2]), "=&v"(fa[3]), "=&v"(fa[4]), "=&v"(fa[5]), "=&v"(fa[6]), "=&v"(fa[7]),
              "=&v"(fa[8]), "=&v"(fa[9]), "=&v"(fb[0]), "=&v"(fb[1]), "=&v"(fb[2]), "=&v"(fb[3]), "=&v"(fb[4]), "=&v"(fb[5]),
              "=&v"(fb[6]), "=&v"(fb[7]), "=&v"(fb[8]), "=&v"(fb[9])
            : "v"(aa), "v"(ab)
            : "memory");
      }
      const lr_f2 Ar = {fa[2], fb[2]}, Br = {fa[3], fb[3]}, Cr = {fa[4], fb[4]};
      const lr_f2 op2 = {fa[5], fb[5]};
      const lr_f2 cr = {fa[6], fb[6]}, cg = {fa[7], fb[7]}, cbl = {fa[8], fb[8]};
      const uint32_t gida = __float_as_uint(fa[9]), gidb = __float_as_uint(fb[9]);
      const lr_f2 dx2 = lr_f2{fa[0], fb[0]} - pxf, dy2 = lr_f2{fa[1], fb[1]} - pyf;
      const lr_f2 hdx2 = dx2 * -0.5f, hdy2 = dy2 * -0.5f;
      const lr_f2 bdx = Br * dx2;
      const lr_f2 pw2 = lr_fma2(Ar * dx2, hdx2, lr_fma2(Cr * dy2, hdy2, lr_f2{-bdx.x, -bdx.y} * dy2));
      const lr_f2 G2 = lr_exp2(pw2);
      const lr_f2 al2 = op2 * G2;
      const float alpha0 = fminf(0.99f, al2.x), alpha1 = fminf(0.99f, al2.y);
      const int k0 = hi - 1 - (int)ja, k1 = hi - 1 - (int)jb;   // list positions (the all-zero slot has opacity 0: never a hit)
      const bool hit0 = (k0 < lastc) & !(pw2.x > 0.f) & !(alpha0 < 1.0f / 255.0f);
      const bool hit1 = (k1 < lastc) & !(pw2.y > 0.f) & !(alpha1 < 1.0f / 255.0f);
      const lr_f2 alpha = {hit0 ? alpha0 : 0.f, hit1 ? alpha1 : 0.f};
      const bool any = __builtin_amdgcn_ballot_w64((alpha.x > 0.f) | (alpha.y > 0.f)) != 0;
      if (!any) continue;
      const lr_f2 G = {hit0 ? G2.x : 0.f, hit1 ? G2.y : 0.f};
      const lr_f2 om = 1.f - alpha;
      lr_f2 rc = {__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
      rc = lr_fma2(lr_fma2(-om, rc, lr_f2{1.f, 1.f}), rc, rc);
      const float Ta = T * rc.x, Tb = Ta * rc.y;
      const lr_f2 T2 = {Ta, Tb};
      T = Tb;
      const lr_f2 w = alpha * T2;
      const float a0r = lr_fma(alpha.x, cr.x, om.x * acc0), a0g = lr_fma(alpha.x, cg.x, om.x * acc1),
                  a0b = lr_fma(alpha.x, cbl.x, om.x * acc2);
      lr_f2 dL_dalpha = lr_fma2(cr - lr_f2{acc0, a0r}, lr_f2{dp0, dp0},
                                lr_fma2(cg - lr_f2{acc1, a0g}, lr_f2{dp1, dp1}, (cbl - lr_f2{acc2, a0b}) * dp2));
      dL_dalpha = lr_fma2(dL_dalpha, T2, -(Tf * rc) * bgdot);
      acc0 = lr_fma(alpha.y, cr.y, om.y * a0r);
      acc1 = lr_fma(alpha.y, cg.y, om.y * a0g);
      acc2 = lr_fma(alpha.y, cbl.y, om.y * a0b);
      const lr_f2 dL_dG = op2 * dL_dalpha;
      const lr_f2 gdx = G * dx2, gdy = G * dy2;
      const lr_f2 dG_ddx = lr_fma2(-Ar, gdx, -(Br * gdy));
      const lr_f2 dG_ddy = lr_fma2(-Cr, gdy, -(Br * gdx));
      const lr_f2 c0 = w * dp0, c1 = w * dp1, c2 = w * dp2, go = G * dL_dalpha;
      const lr_f2 mxs = dL_dG * dG_ddx * sx, mys = dL_dG * dG_ddy * sy;
      const lr_f2 kA = -0.5f * gdx * dx2 * dL_dG, kB = -gdx * dy2 * dL_dG, kC = -0.5f * gdy * dy2 * dL_dG;
      // row reductions, packed: S1 quads = (col r, col b, col g, opacity), S2 quads = (mean x, conic A, mean y, conic B)
      const float s1a = lr_quad_total(lr_row_pair4(lr_row_pair8(c0.x, c1.x), lr_row_pair8(c2.x, go.x)));
      const float s1b = lr_quad_total(lr_row_pair4(lr_row_pair8(c0.y, c1.y), lr_row_pair8(c2.y, go.y)));
      const float s2a = lr_quad_total(lr_row_pair4(lr_row_pair8(mxs.x, mys.x), lr_row_pair8(kA.x, kB.x)));
      const float s2b = lr_quad_total(lr_row_pair4(lr_row_pair8(mxs.y, mys.y), lr_row_pair8(kA.y, kB.y)));
      float s3 = lr_row_pair8(kC.x, kC.y);                      // lanes 0-7: entry 0's conic C, lanes 8-15: entry 1's
      s3 = lr_quad_total(s3 + lr_dpp_perm<0x141>(s3));
      // one instruction per entry: nine lanes of every row, one 64-byte line per row
      const float xa = sel == 0 ? s1a : (sel == 1 ? s2a : s3);
      const float xb = sel == 0 ? s1b : (sel == 1 ? s2b : s3);
      if (!LR_ABLATED(1)) {
        if (on_a && gida != 0xffffffffu) lr_atomic_add_noret(dst + (size_t)gida * LOGRAST_BWD_ROW_FLOATS, xa);
        if (on_b && gidb != 0xffffffffu) lr_atomic_add_noret(dst + (size_t)gidb * LOGRAST_BWD_ROW_FLOATS, xb);
      }
    }
  }
}

// ---- forward, row-split form ------------------------------------------------------------------------------------
// The forward counterpart of lr_blend_bwd_rows_kernel: the wave's four 16-lane rows composite their own 4x4 blocks, each
// taking its own next two relevant entries per pass.  Per pixel the op sequence is the quadrant kernel's (same 2-wide
// power / exp / alpha, entries in list order), so image, final_T, n_contrib and the fork maps stay bit-identical to the
// oracle.  A row stops taking entries when its 16 pixels are saturated; point_weight gets one atomicMax per contributing
// (row, Gaussian) visit (row maximum by four DPP steps), and the Gaussian's accumulator row is cleared by the row's
// first four lanes.
template <bool EXTRAS>
__global__ void __launch_bounds__(256) LR_OCC_FWD_ROWS
lr_blend_fwd_rows_kernel(LrView v, const float4* __restrict__ geom, const uint32_t* __restrict__ state,
                         uint32_t tiles, const uint32_t* __restrict__ plist, uint32_t capacity,
                         float* __restrict__ image, float* __restrict__ final_T, int* __restrict__ n_contrib,
                         int* __restrict__ pid, float* __restrict__ pwp, float* __restrict__ pw,
                         float4* __restrict__ zero_rows, int xcd_mode, int cull, uint32_t* __restrict__ lazy_state,
                         int lazy, uint64_t* __restrict__ masks, uint32_t* __restrict__ hdr_w, int block_test LR_ABLATE_PARAM) {
  __shared__ float4 lr_stage[4][65 * LR_RB_SLOT];
  if (lr_bail(state, capacity)) return;
  if (lazy == 2 && !lazy_state[LR_HDR_OPEN]) return;         // nobody parked (lazy_state: the tile state again, through the pointer these kernels WRITE sorted[] / open[] / the flag with)
  if (masks && blockIdx.x == 0 && threadIdx.x == 0) hdr_w[LR_HDR_MASKS] = LR_MASK_FORM_ROWS;
  const uint32_t tile = lr_tile_of_block(blockIdx.x, tiles, v.gx, v.gy, xcd_mode, state);
  if (tile >= tiles) return;
  const uint32_t* offsets = state + lr_offsets_off(tiles);
  uint32_t beg = offsets[tile];
  uint32_t end = offsets[tile + 1], first;
  bool clamped;
  const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
  // Hit masks for the reverse walk: collected in LDS (64 chunks x 4 block masks per wave) and written out in bursts -- when
  // the buffer is full and after the walk.  A store per chunk, straight from the loop, cost the kernel 60-90 us at 30 M
  // (544 -> 602-632): gfx9 counts loads and stores in ONE counter (vmcnt) and they return out of order with each other, so
  // with a store in flight every wait for a prefetched record becomes vmcnt(0) -- the two-chunk software pipeline of this
  // loop drains once per chunk.  (The quadrant kernel's records come through the scalar cache: measured neutral there.)
  __shared__ uint64_t lr_mbuf[4][LR_MBUF_CHUNKS * 4];
  __shared__ uint32_t lr_wmax_r[4][65];                      // per wave and chunk: the running maximum of alpha T of every entry (+ slot 64: rows without work)
  uint32_t* const wmx = lr_wmax_r[wq];
  if (EXTRAS) { wmx[lane] = 0u; if (lane == 0) wmx[64] = 0u; }
  uint32_t id_prev = 0xffffffffu;                            // the ids of the chunk whose commit is pending (lr_fwd_commit_chunk)
  uint64_t* const mslot = masks ? masks + 16 * lr_mask_slot(beg, tile) + 4 * wq : nullptr;
  uint32_t mcount = 0;                                       // chunks waiting in lr_mbuf[wq] (wave-uniform)
  uint32_t mfirst = 0;                                       // ... the first of them (chunk index inside the tile's list)
  auto flush_masks = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!LR_ABLATED(8))
    for (uint32_t i = (uint32_t)lane; i < mcount * 4u; i += 64u)   // lane -> (chunk i / 4, block i % 4): 32 adjacent bytes per chunk
      mslot[16 * (size_t)(mfirst + (i >> 2)) + (i & 3u)] = lr_mbuf[wq][i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    mfirst += mcount;
    mcount = 0;
  };
  if (!lr_lazy_range(lazy_state + lr_sorted_off(tiles), tiles, lazy, tile, wq, beg, end, first, clamped)) return;
  const int row = lane >> 4, li = lane & 15;
  const int tx = tile % (uint32_t)v.gx, ty = tile / (uint32_t)v.gx;
  const int qx0 = tx * 16 + (wq & 1) * 8, qy0 = ty * 16 + (wq >> 1) * 8;
  const int px = qx0 + (row & 1) * 4 + (li & 3), py = qy0 + (row >> 1) * 4 + (li >> 2);
  const float pxf = (float)px, pyf = (float)py;
  const bool inside = (px < v.W) && (py < v.H);
  const size_t pix = inside ? (size_t)py * v.W + px : 0;
  bool done = !inside;
  float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, wmax = 0.f;
  int wid = -1, last = 0;
  if (lazy == 2) {                                           // second pass: this wave parked at list position `first`
    if (inside) lr_lazy_resume<EXTRAS>(v, pix, done, T, C0, C1, C2, last, wid, wmax, image, final_T, n_contrib, pid, pwp);
    beg += first;
  }
  float4* const stage = lr_stage[wq];
  if (lane < LR_RB_SLOT) stage[64 * LR_RB_SLOT + lane] = float4{0.f, 0.f, 0.f, 0.f};
  if (lane == 2) stage[64 * LR_RB_SLOT + 2] = float4{0.f, __uint_as_float(0xffffffffu), 0.f, 0.f};
  const uint32_t shift16 = 16u * (uint32_t)row;
  const float bx[2] = {(float)qx0, (float)(qx0 + 4)}, by[2] = {(float)qy0, (float)(qy0 + 4)};
  const uint64_t rowbits = 0xffffull;

  const uint32_t nchunks = (end - beg + 63u) >> 6;
  if (end == beg) {                                          // an empty list (the loads below are unconditional: plist[beg] must exist)
    if (inside && lazy != 2) {
      const size_t plane = (size_t)v.W * v.H;
      final_T[pix] = 1.f; n_contrib[pix] = 0;
      image[pix] = v.bg[0]; image[plane + pix] = v.bg[1]; image[2 * plane + pix] = v.bg[2];   // (fma(1, bg, 0) = bg)
      if (EXTRAS) { pid[pix] = -1; pwp[pix] = 0.f; }
    }
    return;
  }
  // Unconditional loads, the gather of chunk ch + 1 in front of the ids of chunk ch + 2 (see lr_blend_bwd_rows_kernel: as
  // exec-masked branches they drew an s_waitcnt vmcnt(0) right behind the id load of every chunk)
  auto load_id = [&](uint32_t c) -> uint32_t {
    const uint32_t idx = beg + c * 64u + (uint32_t)lane;
    const bool ok = c < nchunks && idx < end;
    const uint32_t got = plist[ok ? idx : beg];
    return ok ? got : 0xffffffffu;
  };
  auto gather = [&](uint32_t idv, float4& q0, float4& q1, float& qc) {
    const float4* rp = geom + LR_REC_QUADS * (size_t)(idv != 0xffffffffu ? idv : 0u);   // (record 0: inside the buffer, never used)
    q0 = rp[0]; q1 = rp[1]; qc = reinterpret_cast<const float*>(rp)[8];
  };
  uint32_t id_n = load_id(0), id_nn = load_id(1);
  float4 g0_n, g1_n;
  float cb_n;
  gather(id_n, g0_n, g1_n, cb_n);
  LR_LANDED(g0_n, g1_n, cb_n, id_nn);                        // (see lr_blend_bwd_rows_kernel: the loop's entry state = its steady state)

  for (uint32_t ch = 0; ch < nchunks; ch++) {
    if (__all(done)) break;
    lr_fwd_commit_chunk<EXTRAS>(wmx, lane, id_prev, pw, zero_rows);   // the previous chunk's, in front of this chunk's loads
    const uint32_t id = id_n;
    id_prev = id;
    const float4 g0 = g0_n, g1 = g1_n;
    const float cb = cb_n;
    id_n = id_nn;
    gather(id_n, g0_n, g1_n, cb_n);
    id_nn = load_id(ch + 2);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    stage[lane * LR_RB_SLOT + 0] = g0;
    stage[lane * LR_RB_SLOT + 1] = g1;
    stage[lane * LR_RB_SLOT + 2] = float4{cb, __uint_as_float(id), 0.f, 0.f};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool valid = id != 0xffffffffu;
    bool r0 = valid, r1 = valid, r2 = valid, r3 = valid;
    if (cull) {
      const LrSupport sp = lr_support_prepare(g0.x, g0.y, g0.z, g0.w, g1.x, g1.y);
      if (block_test) {
        // the four 4x4 blocks two at a time (lr_support_box2: the decisions of four lr_support_box calls, packed arithmetic)
        bool k0, k1, k2, k3;
        const lr_f2 X0 = {bx[0], bx[1]}, X1 = {bx[0] + 3.f, bx[1] + 3.f};
        lr_support_box2(sp, X0, X1, lr_f2{by[0], by[0]}, lr_f2{by[0] + 3.f, by[0] + 3.f}, k0, k1);
        lr_support_box2(sp, X0, X1, lr_f2{by[1], by[1]}, lr_f2{by[1] + 3.f, by[1] + 3.f}, k2, k3);
        r0 = r0 && k0; r1 = r1 && k1; r2 = r2 && k2; r3 = r3 && k3;
      } else {            // LOGRAST_FWD_BLOCK_TEST=0: exact test for the quadrant, the support's bounding box against each block
        const bool q = lr_support_box(sp, bx[0], bx[0] + 7.f, by[0], by[0] + 7.f);
        const bool bb = sp.mode == 2;
        const bool x_lo = !bb || (sp.mx - sp.ex <= bx[0] + 3.f), x_hi = !bb || (sp.mx + sp.ex >= bx[1]);
        const bool y_lo = !bb || (sp.my - sp.ey <= by[0] + 3.f), y_hi = !bb || (sp.my + sp.ey >= by[1]);
        r0 = r0 && q && x_lo && y_lo;
        r1 = r1 && q && x_hi && y_lo;
        r2 = r2 && q && x_lo && y_hi;
        r3 = r3 && q && x_hi && y_hi;
      }
    }
    uint64_t mrow = lr_row_mask(row, __ballot(r0), __ballot(r1), __ballot(r2), __ballot(r3));   // this lane's row's hit mask
    const int pos0 = (int)(first + ch * 64u);
    if (mslot) {                                              // for the reverse walk (hit masks, above)
      if (mcount == 0) mfirst = (uint32_t)pos0 >> 6;
      if (li == 0 && !LR_ABLATED(16)) lr_mbuf[wq][mcount * 4u + (uint32_t)row] = mrow;
      if (++mcount == LR_MBUF_CHUNKS) flush_masks();
    }
    if (LR_ABLATED(4)) mrow = 0ull;   // experiment builds: the chunk prologue alone (gathers, staging, support tests), every list to its end
    while (true) {
      // a row whose 16 pixels are all saturated takes no more entries
      const uint64_t dm = __ballot(done);
      if ((uint32_t)((dm >> shift16) & rowbits) == (uint32_t)rowbits) mrow = 0ull;
      if (__builtin_amdgcn_ballot_w64(mrow != 0ull) == 0) break;
      const uint32_t ja = lr_take_bit(mrow), jb = lr_take_bit(mrow);
      const float4* sa = stage + ja * LR_RB_SLOT;
      const float4* sb = stage + jb * LR_RB_SLOT;
      const float4 a0 = sa[0], a1 = sa[1], a2 = sa[2], b0 = sb[0], b1 = sb[1], b2 = sb[2];
      const int gida = (int)__float_as_uint(a2.y), gidb = (int)__float_as_uint(b2.y);
      const float op0 = a1.y, op1 = b1.y;
      const lr_f2 dx2 = lr_f2{a0.x, b0.x} - pxf, dy2 = lr_f2{a0.y, b0.y} - pyf;
      const lr_f2 hdx2 = dx2 * -0.5f, hdy2 = dy2 * -0.5f;
      const lr_f2 bdx = lr_f2{a0.w, b0.w} * dx2;
      const lr_f2 pw2 = lr_fma2(lr_f2{a0.z, b0.z} * dx2, hdx2,
                                lr_fma2(lr_f2{a1.x, b1.x} * dy2, hdy2, lr_f2{-bdx.x, -bdx.y} * dy2));
      const lr_f2 al2 = lr_f2{op0, op1} * lr_exp2(pw2);
      const float alpha0 = fminf(0.99f, al2.x), alpha1 = fminf(0.99f, al2.y);
      // entry a, then entry b, in list order (the all-zero slot of a row without work has opacity 0: never ok)
      const bool ok0 = !done & !(pw2.x > 0.f) & !(alpha0 < 1.0f / 255.0f);
      const float test0 = T * (1.f - alpha0);
      const bool stop0 = ok0 & (test0 < 0.0001f);
      const bool acc0 = ok0 & !stop0;
      const float w0 = acc0 ? alpha0 * T : 0.f;
      T = acc0 ? test0 : T;
      last = acc0 ? pos0 + (int)ja + 1 : last;
      done = done | stop0;
      const bool ok1 = !done & !(pw2.y > 0.f) & !(alpha1 < 1.0f / 255.0f);
      const float test1 = T * (1.f - alpha1);
      const bool stop1 = ok1 & (test1 < 0.0001f);
      const bool acc1 = ok1 & !stop1;
      const float w1 = acc1 ? alpha1 * T : 0.f;
      T = acc1 ? test1 : T;
      last = acc1 ? pos0 + (int)jb + 1 : last;
      done = done | stop1;
      const bool hit = __builtin_amdgcn_ballot_w64((w0 > 0.f) | (w1 > 0.f)) != 0;
      if (!hit) continue;
      if (acc0) { C0 = lr_fma(a1.z, w0, C0); C1 = lr_fma(a1.w, w0, C1); C2 = lr_fma(a2.x, w0, C2); }
      if (acc1) { C0 = lr_fma(b1.z, w1, C0); C1 = lr_fma(b1.w, w1, C1); C2 = lr_fma(b2.x, w1, C2); }
      if (EXTRAS) {
        if (w0 > wmax) { wmax = w0; wid = gida; }
        if (w1 > wmax) { wmax = w1; wid = gidb; }
        // row maxima (non-negative floats order as unsigned): every lane of the row ends up with the row's maximum
        uint32_t ma = __float_as_uint(w0), mb = __float_as_uint(w1);
#define LR_RMAX(x, CTRL) x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true))
        LR_RMAX(ma, 0x140); LR_RMAX(mb, 0x140);
        LR_RMAX(ma, 0x141); LR_RMAX(mb, 0x141);
        LR_RMAX(ma, 0xB1); LR_RMAX(mb, 0xB1);
        LR_RMAX(ma, 0x4E); LR_RMAX(mb, 0x4E);
#undef LR_RMAX
        // (LDS, one lane per row and entry; point_weight and the row clears leave once per chunk: lr_fwd_commit_chunk)
        if (li == 0 && ma != 0u && !LR_ABLATED(1)) atomicMax(&wmx[ja], ma);
        if (li == 0 && mb != 0u && !LR_ABLATED(1)) atomicMax(&wmx[jb], mb);
      }
    }
  }
  if (mslot && mcount) flush_masks();
  lr_fwd_commit_chunk<EXTRAS>(wmx, lane, id_prev, pw, zero_rows);   // the last chunk's
  if (clamped && !__all(done)) {                             // out of ordered entries with a pixel open: to be continued
    if (lane == 0) { atomicOr(lazy_state + lr_sorted_off(tiles) + tiles + tile, 1u << wq); atomicOr(lazy_state + LR_HDR_OPEN, 1u); }   // open[tile], header
    if (inside) lr_lazy_park<EXTRAS>(v, pix, done, T, C0, C1, C2, last, wid, wmax, image, final_T, n_contrib, pid, pwp);
    return;
  }

  if (inside) {
    const size_t plane = (size_t)v.W * v.H;
    final_T[pix] = T;
    n_contrib[pix] = last;
    image[pix] = lr_fma(T, v.bg[0], C0);
    image[plane + pix] = lr_fma(T, v.bg[1], C1);
    image[2 * plane + pix] = lr_fma(T, v.bg[2], C2);
    if (EXTRAS) { pid[pix] = wid; pwp[pix] = wmax; }
  }
}

// Which form the forward's compositing launches for this view (1 = row-split, 2 = quadrant): lr_launch_blend_fwd's rule,
// also behind lograst_forward_form (the caller of lograst_backward passes it back as lograst_view.hit_mask_form).
int lr_blend_fwd_form(const LrView& v) {
  LR_KNOB(rows_knob, "LOGRAST_FWD_ROWS", 2);
  const int rows = rows_knob != 2 ? rows_knob : (v.walk_form == LOGRAST_FORM_ROWS ? 1 : 0);   // no hint: quadrant
  return rows ? (int)LR_MASK_FORM_ROWS : (int)LR_MASK_FORM_QUAD;
}

void lr_launch_blend_fwd(const LrView& v, const void* geom, const uint32_t* state, uint32_t tiles,
                         const uint32_t* plist, uint32_t capacity, float* image, float* final_T, int* n_contrib,
                         int* pid, float* pwp, float* pw, float* zero_conic, int big_input, int lazy, uint64_t* masks,
                         hipStream_t s) {
  uint32_t* const hdr_w = const_cast<uint32_t*>(state);     // (header word LR_HDR_MASKS: which form left hit masks)
  uint32_t* const lazy_state = lazy ? const_cast<uint32_t*>(state) : nullptr;   // (the tile state once more, writable: open[] and header word LR_HDR_OPEN are all a compositing kernel writes there)
  LR_KNOB(xcd_knob, "LOGRAST_XCD_MODE", 3);
  int xcd_mode = xcd_knob;
  static const int cull = LR_EXPERIMENT_INT("LOGRAST_CULL", 1);   // experiment builds: 0 = no per-quadrant support test
  static const size_t lds_fwd = (size_t)LR_EXPERIMENT_INT("LOGRAST_BLEND_FWD_LDS_KB", 0) * 1024;   // experiment builds: occupancy cap
  // LOGRAST_FWD_ROWS: 1 = row-split form (lr_blend_fwd_rows_kernel), 0 = one quadrant per wave, 2 (default) = the caller's
  // hint (lograst_view.walk_form), quadrant without one.  Measured, MI355X: 30 M tiny splats 706 -> 658 us (random
  // opacities 1267 -> 1188); C2's 1 M 174 -> 193; a tree-ordered heavy-tailed view 278 -> 347.
#ifdef LR_EXPERIMENTS
  static const int fwd_ablate = lr_env_int("LOGRAST_FWD_ABLATE", 0);   // timing experiments (row-split form): 1 no point_weight atomics, 2 no row clears
#endif
  const int rows = lr_blend_fwd_form(v) == (int)LR_MASK_FORM_ROWS;
  LR_KNOB(fwd_block_test, "LOGRAST_FWD_BLOCK_TEST", 1);
  if (!cull) masks = nullptr;                                // (experiment builds without support tests: nothing to hand over)
  uint32_t grid = lr_blend_grid(tiles, v.gx, v.gy, xcd_mode);
  if (lazy == 2) {   // only streamed lists can be open: in the scan's longest-first order they sit in front of every shorter one
    xcd_mode = 3;
    grid = min(tiles, capacity / (uint32_t)LR_LONG_LIST + 1u);
  }
  const float4* g4 = reinterpret_cast<const float4*>(geom);
  float4* z4 = reinterpret_cast<float4*>(zero_conic);
  if (lazy != 2) lr_prof_begin(LRK_BLEND_FWD, s);           // (the second pass is timed by its caller, with the sort of the tails)
  if (rows) {
    if (v.extras)
      hipLaunchKernelGGL(lr_blend_fwd_rows_kernel<true>, dim3(grid), dim3(256), lds_fwd, s, v, g4, state, tiles, plist, capacity,
                         image, final_T, n_contrib, pid, pwp, pw, z4, xcd_mode, cull, lazy_state, lazy, masks, hdr_w, fwd_block_test LR_ABLATE_PASS(fwd_ablate));
    else
      hipLaunchKernelGGL(lr_blend_fwd_rows_kernel<false>, dim3(grid), dim3(256), lds_fwd, s, v, g4, state, tiles, plist, capacity,
                         image, final_T, n_contrib, pid, pwp, pw, z4, xcd_mode, cull, lazy_state, lazy, masks, hdr_w, fwd_block_test LR_ABLATE_PASS(fwd_ablate));
  } else {
    if (v.extras)
      hipLaunchKernelGGL(lr_blend_fwd_kernel<true>, dim3(grid), dim3(256), lds_fwd, s, v, g4, state, tiles, plist, capacity,
                         image, final_T, n_contrib, pid, pwp, pw, z4, xcd_mode, cull, lazy_state, lazy, masks, hdr_w);
    else
      hipLaunchKernelGGL(lr_blend_fwd_kernel<false>, dim3(grid), dim3(256), lds_fwd, s, v, g4, state, tiles, plist, capacity,
                         image, final_T, n_contrib, pid, pwp, pw, z4, xcd_mode, cull, lazy_state, lazy, masks, hdr_w);
  }
  if (lazy != 2) lr_prof_end(LRK_BLEND_FWD, s);
}

void lr_launch_blend_bwd(const LrView& v, const void* geom, const uint32_t* state, uint32_t tiles,
                         const uint32_t* plist, uint32_t capacity, const float* final_T, const int* n_contrib,
                         const float* dL_dimage, float* acc_rows, int big_input, const uint64_t* masks, hipStream_t s) {
  LR_KNOB(xcd_mode, "LOGRAST_XCD_MODE", 3);
  static const int cull = LR_EXPERIMENT_INT("LOGRAST_CULL", 1);   // experiment builds: 0 = no per-quadrant support test
  static const size_t lds_bwd = (size_t)LR_EXPERIMENT_INT("LOGRAST_BLEND_BWD_LDS_KB", 0) * 1024;   // experiment builds: occupancy cap
  uint32_t grid = lr_blend_grid(tiles, v.gx, v.gy, xcd_mode);
  // LOGRAST_BWD_ROWS=1: the row-split form (the four 16-lane rows of a wave walk their own 4x4 blocks); 0: one
  // (Gaussian, quadrant) pair per visit; 2 (default): the caller's hint (lograst_view.walk_form: row-split for views of
  // tiny splats, few tile instances per Gaussian), else row-split on large inputs.  Measured, MI355X: 30 M tiny splats
  // 949 -> 700 us, with random opacities 1768 -> 1259; C2's 1 M 292 -> 307; a tree-ordered heavy-tailed view 448 -> 579.  LOGRAST_BWD_ABLATE (timing experiments): 1 = no atomics in the row-split form.
  LR_KNOB(rows_knob, "LOGRAST_BWD_ROWS", 2);
  const int rows = rows_knob != 2 ? rows_knob
                   : (v.walk_form == LOGRAST_FORM_ROWS ? 1 : (v.walk_form == LOGRAST_FORM_QUADRANT ? 0 : (big_input ? 1 : 0)));
#ifdef LR_EXPERIMENTS
  static const int ablate = lr_env_int("LOGRAST_BWD_ABLATE", 0);   // timing experiments: 1 = no atomics in the row-split form
#endif
  LR_KNOB(block_test, "LOGRAST_BWD_BLOCK_TEST", 1);
  lr_prof_begin(LRK_BLEND_BWD, s);
  // the forward's hit masks serve a reverse walk of the SAME form only (v.mask_form: what the caller says its forward launched)
  const bool use_masks = masks != nullptr && cull && v.mask_form == (rows ? (int)LR_MASK_FORM_ROWS : (int)LR_MASK_FORM_QUAD);
  const float4* g4 = reinterpret_cast<const float4*>(geom);
  if (rows && use_masks)
    hipLaunchKernelGGL(lr_blend_bwd_rows_kernel<true>, dim3(grid), dim3(256), lds_bwd, s, v, g4, state, tiles, plist, capacity,
                       final_T, n_contrib, dL_dimage, acc_rows, xcd_mode, cull, block_test, masks LR_ABLATE_PASS(ablate));
  else if (rows)
    hipLaunchKernelGGL(lr_blend_bwd_rows_kernel<false>, dim3(grid), dim3(256), lds_bwd, s, v, g4, state, tiles, plist, capacity,
                       final_T, n_contrib, dL_dimage, acc_rows, xcd_mode, cull, block_test, masks LR_ABLATE_PASS(ablate));
  else if (use_masks)
    hipLaunchKernelGGL(lr_blend_bwd_kernel<true>, dim3(grid), dim3(256), lds_bwd, s, v, g4, state, tiles, plist, capacity,
                       final_T, n_contrib, dL_dimage, acc_rows, xcd_mode, cull, masks);
  else
    hipLaunchKernelGGL(lr_blend_bwd_kernel<false>, dim3(grid), dim3(256), lds_bwd, s, v, g4, state, tiles, plist, capacity,
                       final_T, n_contrib, dL_dimage, acc_rows, xcd_mode, cull, masks);
  lr_prof_end(LRK_BLEND_BWD, s);
}

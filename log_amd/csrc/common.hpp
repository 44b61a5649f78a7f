// common.hpp -- shared device helpers for liblograst (gfx950 / CDNA4 only; wave = 64).
//
// Numerics contract: compiled with -ffp-contract=off; every fused multiply-add is an explicit
// __builtin_fmaf.  All operations feeding integer outputs (radii, tile rects, per-tile order,
// n_contrib, point_id_pixel) follow one fixed fp32 op sequence so that the CPU oracle (test
// infrastructure, never linked here) can reproduce them bit-for-bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#include "../../include/lograst.h"

#define LR_WAVE 64
#define LR_TILE 16

// ---- tile_state layout (uint32 words) ------------------------------------------------------------
// [0] num_instances  [1] overflow flag  [2] longest tile list  [3] reserved
// [4] tile instances of the plain rect rule (before the support cull; reporting only)
// [5] support cull applied by the projection kernel (0/1)  [6] projection batch size (0 = unbatched)
// [7] some rect was deferred to lr_count_huge_kernel  [8] band views: fill-record slots per projection workgroup
// [9] the long-list sort left lists at their first window (sorted[] / open[] below are valid)  [10] a compositing wave parked at the end of an ordered part  [11..15] reserved
// then per-tile counter arrays, one counter every S = LR_CTR_STRIDE words (64 B apart: the atomic targets
// spread over the memory channels instead of 8160 counters sharing 32 KB); header, ranked and big are
// contiguous so that ONE memset prepares a forward:
//   ranked[T*S]   instances of Gaussians touching <= LR_RANKED_TILES tiles; the returning atomic that counts
//                 them also hands each instance its slot inside the tile (stored in the record's q3;
//                 0xffffffff = this tile of the rect was dropped by the support cull)
//   big[T*S]      instances of larger Gaussians (counted only)
// then offsets[Tp]: exclusive offsets (T+1 entries)                                   -- read by sort/blend
//   cursor[T*S]   fill cursor for the big instances, initialised to offset + ranked
// then order[T]: tile ids by descending list length (longest-first dispatch order for the blend kernels)
#ifndef LR_CTR_STRIDE
#define LR_CTR_STRIDE 16
#endif
#define LR_RANKED_TILES 4
#define LR_HDR_WORDS 16
#define LR_HDR_NUM 0
#define LR_HDR_OVERFLOW 1
#define LR_HDR_MAXLEN 2
#define LR_HDR_BATCH 6  // Gaussians per projection batch (0 = unbatched kernel: slots in q3 are absolute)
#define LR_HDR_CULL 5  // 1 if lr_project_kernel applied the support cull (the fill kernel must repeat it)
#define LR_HDR_RECT 4  // tile instances of the plain rect rule (what the reference would sort), before the support cull
#define LR_HDR_SPARSE 3  // band view (lr_project_band_kernel): only Gaussians with a rect have a record; their fill records are compacted per projection workgroup:
#define LR_HDR_SPAN 8    // workgroup w owns fill-record slots [w * span, w * span + survcount[w]), the Gaussians' indices behind the fill records
#define LR_HDR_HUGE 7  // batched projection: some workgroup deferred a rect to lr_count_huge_kernel (else that kernel returns at once)
#define LR_HDR_LAZY 9  // the per-tile sort ordered only the first window of the streamed lists (sorted[] below is valid)
#define LR_HDR_KEYS_LO 11 // fingerprint of the key buffer the fill wrote into (pointer, capacity): lograst_finish_lists refuses any other
#define LR_HDR_KEYS_HI 12
#define LR_HDR_KEYS_CAP 13
#define LR_HDR_MASKS 14 // which compositing form left its per-chunk support ballots in the caller's hit-mask buffer (0 none, 1 row-split, 2 quadrant: blend.hip)
#define LR_HDR_OPEN 10 // some compositing wave parked at the end of an ordered part (else the second sort / compositing pair returns at once)
#define LR_SORT_BLOCK 8192  // keys one workgroup sorts in LDS
#define LR_LONG_LIST 4096   // longer lists are sorted with their keys streamed from memory (shorter ones: LDS-resident)
#define LR_REC_QUADS 4  // float4 per projected record (64 B)
__host__ __device__ inline uint32_t lr_tpad(uint32_t tiles) { return (tiles + 1 + 15u) & ~15u; }
// header | ranked | big are contiguous: one memset clears everything a forward needs zeroed
__host__ __device__ inline uint32_t lr_ranked_off(uint32_t tiles) { (void)tiles; return LR_HDR_WORDS; }
__host__ __device__ inline uint32_t lr_big_off(uint32_t tiles) { return lr_ranked_off(tiles) + tiles * LR_CTR_STRIDE; }
__host__ __device__ inline uint32_t lr_offsets_off(uint32_t tiles) { return lr_big_off(tiles) + tiles * LR_CTR_STRIDE; }
__host__ __device__ inline uint32_t lr_cursor_off(uint32_t tiles) { return lr_offsets_off(tiles) + lr_tpad(tiles); }
__host__ __device__ inline uint32_t lr_order_off(uint32_t tiles) { return lr_cursor_off(tiles) + tiles * LR_CTR_STRIDE; }
// sorted[T] | open[T]: once the fill is done its cursors are dead, and the first 2 T words of their array say how much of
// each STREAMED list (more than LR_LONG_LIST keys) is in final order -- its first sorted[t] positions -- and which of the
// tile's four compositing waves ran out of ordered entries with a pixel still open (open[t], one bit per wave).  The walk
// of a view almost never leaves a long list's first window (30 M Gaussians @1080p: 2116 lists of 19.6 K keys each, the
// deepest pixel of any of them stops after 1.4 K entries, 2.8 K with random opacities: tools/walk_depth_probe.py), so the
// sort orders that window only (sort.hip, lazy = 1); a compositing wave that gets to its end with a pixel open parks its
// pixels' state in their outputs and sets its bit, and a second, normally idle pair of launches orders the rest of exactly
// those lists and lets exactly those waves go on where they stopped (blend.hip: lr_lazy_range).
__host__ __device__ inline uint32_t lr_sorted_off(uint32_t tiles) { return lr_cursor_off(tiles); }
// then basetab[batches][T]: start of every projection batch's reservation inside each tile's ranked range
__host__ __device__ inline uint32_t lr_basetab_off(uint32_t tiles) { return lr_order_off(tiles) + lr_tpad(tiles); }
// then hugemask[batches][LR_HUGE_WORDS]: which 256-Gaussian chunks of the batch hold a Gaussian that left its (more than
// LR_COOP_TILES tile) rect to lr_count_huge_kernel -- bit c of the batch's 128 = chunk c (a batch is at most 32768
// Gaussians).  (Rounds 1-4 kept one COUNT per batch: in a scene whose few large splats are spread evenly -- any trained
// model in storage order -- every batch holds one and the counting kernel then walked all N fill records, clearing and
// flushing its 8160 LDS tile counters once per 256 Gaussians: 459 us at 30 M for ~1000 rects.)
#define LR_HUGE_WORDS 4
__host__ __device__ inline size_t lr_hugecount_off(uint32_t tiles, uint32_t batches) {
  return (size_t)lr_basetab_off(tiles) + (size_t)batches * tiles;
}
__host__ __device__ inline uint32_t lr_hugemask_words(uint32_t batches) { return (LR_HUGE_WORDS * batches + 15u) & ~15u; }
// then survcount[batches] (band views: survivors of projection workgroup w at [w], lr_project_band_kernel)
__host__ __device__ inline size_t lr_survcount_off(uint32_t tiles, uint32_t batches) {
  return lr_hugecount_off(tiles, batches) + lr_hugemask_words(batches);
}
__host__ __device__ inline size_t lr_state_words(uint32_t tiles, uint32_t batches) {
  return lr_survcount_off(tiles, batches) + ((batches + 15u) & ~15u);
}
#define LR_COOP_TILES 16   // rects above this many tiles are expanded by a whole wave (lanes = tiles), not by their lane
#define LR_HUGE_CHUNK 256  // Gaussians per workgroup of lr_count_huge_kernel (a chunk full of 81-tile rects is a serial walk per wave: 2048 ran 0.36 ms on the tree-ordered view, 512 0.14, 256 0.093)
#define LR_BATCH_THREADS 1024
// lr_fill_staged_kernel (project.hip; the host's choice of its template parameter: api.hip): threads per workgroup = Gaussians
// per thread-slot (measured at 30 M, two Gaussians per thread: 1024 threads 327 us, 512: 361-370, 256: 445), and the largest
// tile grid whose slot-table row the workgroup stages in LDS
#ifndef LR_FILL_STAGED_ROWS
#define LR_FILL_STAGED_ROWS 1024
#endif
#define LR_FILL_STAGED_MAX_TILES 12288      // 48 KB of LDS
#define LR_BATCH_MAX_TILES 40000  // 4 B x tiles of LDS counters must fit one workgroup (160 KB): up to 3840x2160
#define LR_BATCH_LDS_BYTES (160 * 1024 - 512)  // dynamic LDS a projection workgroup may use (counter planes)

// Device-side view (kernel argument, by value).
struct LrView {
  int W, H, gx, gy;
  int ty0, ty1;   // tile rows this call owns (image split across GPUs): rects are clipped to [ty0, ty1)
  float tanfovx, tanfovy, fx, fy, scale_modifier;
  int filter_mode, ndc_cull, extras;
  int walk_form;   // LOGRAST_FORM_*
  const float* view;
  const float* proj;
  const float* bg;
  const float* cov3d;   // precomputed world-space covariances (n x 6) instead of scales + rotations, or nullptr
  float* g_cov3d;       // backward: dL/dcov3D (n x 6) when cov3d is set
  uint64_t* masks;      // hit masks: written by the forward's compositing, read by the reverse walk (blend.hip), or nullptr
  uint64_t mask_words;
  int mask_form;        // backward: which compositing form of the forward wrote `masks` (0 unknown / none, 1 row-split, 2 quadrant)
};

#define LR_DEV __device__ __forceinline__

// The view's matrices (device pointers in lograst_view) never change while a kernel runs.  Read through the constant
// address space they are scalar loads (s_load_dword: the scalar cache, counted by lgkmcnt) wherever the compiler needs
// them; read through the flat pointer, a kernel with enough stores in it gets VECTOR loads of these uniform addresses,
// each followed by s_waitcnt vmcnt(0) -- which also waits for the prefetch of the next iteration's inputs.
typedef const float __attribute__((address_space(4))) lr_cfloat;
LR_DEV const lr_cfloat* lr_uniform(const float* p) { return (const lr_cfloat*)p; }

LR_DEV float lr_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
LR_DEV float lr_dot3p(float a0, float a1, float a2, float b0, float b1, float b2, float c) {
  return lr_fma(a0, b0, lr_fma(a1, b1, lr_fma(a2, b2, c)));
}
LR_DEV float lr_dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
  return lr_fma(a0, b0, lr_fma(a1, b1, a2 * b2));
}

// exp(x), x <= 0: 2^(x*log2e) with a degree-5 polynomial on [-0.5, 0.5] (max rel err 2.1e-7).  A fixed
// op sequence instead of v_exp_f32 so that forward decisions (alpha floor, T stop, arg-max) are
// reproducible on the host.
LR_DEV float lr_exp(float x) {
  float t = x * 1.44269504088896341f;
  t = fmaxf(t, -125.0f);
  float n = rintf(t);
  float f = t - n;
  float p = 0x1.5c08e4p-10f;
  p = lr_fma(p, f, 0x1.3d0c52p-7f);
  p = lr_fma(p, f, 0x1.c6b6e4p-5f);
  p = lr_fma(p, f, 0x1.ebf918p-3f);
  p = lr_fma(p, f, 0x1.62e428p-1f);
  p = lr_fma(p, f, 0x1.000002p+0f);
  int ni = (int)n;
  return __uint_as_float(__float_as_uint(p) + ((uint32_t)ni << 23));
}

// power = -0.5 (A dx^2 + C dy^2) - B dx dy with hA = -0.5 A, hC = -0.5 C, nB = -B (exact scalings): the reference op
// sequence shared with oracle/lograst_oracle.c:ora_power.
LR_DEV float lr_power(float hA, float nB, float hC, float dx, float dy) {
  return lr_fma(hA * dx, dx, lr_fma(hC * dy, dy, (nB * dx) * dy));
}

// Two-wide versions (one list entry per half).  Element for element these are the SAME IEEE op sequences as
// lr_power / lr_exp -- v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 round each half exactly like the scalar
// instruction -- so results stay bit-identical to the oracle while the VALU issues half as many instructions
// for the multiply/add/fma part.
typedef float lr_f2 __attribute__((ext_vector_type(2)));
typedef int lr_i2 __attribute__((ext_vector_type(2)));
typedef unsigned int lr_u2 __attribute__((ext_vector_type(2)));
LR_DEV lr_f2 lr_fma2(lr_f2 a, lr_f2 b, lr_f2 c) { return __builtin_elementwise_fma(a, b, c); }
// (the blend kernels evaluate lr_power with the conic straight from the record and the -0.5 / sign scalings moved to
// the pixel side -- fma(A*dx, -0.5*dx, .) rounds the same real number as fma((-0.5*A)*dx, dx, .) -- see blend.hip)
LR_DEV lr_f2 lr_exp2(lr_f2 x) {
  lr_f2 t = x * 1.44269504088896341f;
  t = lr_f2{fmaxf(t.x, -125.0f), fmaxf(t.y, -125.0f)};
  const lr_f2 n = lr_f2{rintf(t.x), rintf(t.y)};
  const lr_f2 f = t - n;
  lr_f2 p = lr_f2{0x1.5c08e4p-10f, 0x1.5c08e4p-10f};
  p = lr_fma2(p, f, lr_f2{0x1.3d0c52p-7f, 0x1.3d0c52p-7f});
  p = lr_fma2(p, f, lr_f2{0x1.c6b6e4p-5f, 0x1.c6b6e4p-5f});
  p = lr_fma2(p, f, lr_f2{0x1.ebf918p-3f, 0x1.ebf918p-3f});
  p = lr_fma2(p, f, lr_f2{0x1.62e428p-1f, 0x1.62e428p-1f});
  p = lr_fma2(p, f, lr_f2{0x1.000002p+0f, 0x1.000002p+0f});
  const int n0 = (int)n.x, n1 = (int)n.y;
  return lr_f2{__uint_as_float(__float_as_uint(p.x) + ((uint32_t)n0 << 23)),
               __uint_as_float(__float_as_uint(p.y) + ((uint32_t)n1 << 23))};
}

// cov3D = R diag(s^2) R^T; quaternion (r,x,y,z) is NOT normalised
// (/root/reference/LoG/cuda/compute_radius_kernel.cu:28-58, :36).
LR_DEV void lr_cov3d(const float s[3], const float q[4], float R[9], float Sg[6]) {
  float r = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1.f - 2.f * lr_fma(y, y, z * z); R[1] = 2.f * lr_fma(x, y, -(r * z)); R[2] = 2.f * lr_fma(x, z, r * y);
  R[3] = 2.f * lr_fma(x, y, r * z); R[4] = 1.f - 2.f * lr_fma(x, x, z * z); R[5] = 2.f * lr_fma(y, z, -(r * x));
  R[6] = 2.f * lr_fma(x, z, -(r * y)); R[7] = 2.f * lr_fma(y, z, r * x); R[8] = 1.f - 2.f * lr_fma(x, x, y * y);
  float M[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int k = 0; k < 3; k++) M[3 * i + k] = R[3 * i + k] * s[k];
  Sg[0] = lr_dot3(M[0], M[1], M[2], M[0], M[1], M[2]);
  Sg[1] = lr_dot3(M[0], M[1], M[2], M[3], M[4], M[5]);
  Sg[2] = lr_dot3(M[0], M[1], M[2], M[6], M[7], M[8]);
  Sg[3] = lr_dot3(M[3], M[4], M[5], M[3], M[4], M[5]);
  Sg[4] = lr_dot3(M[3], M[4], M[5], M[6], M[7], M[8]);
  Sg[5] = lr_dot3(M[6], M[7], M[8], M[6], M[7], M[8]);
}

struct LrEwa {
  float t[3];
  float ux, uy;
  int cx, cy;
  float txc, tyc;
  float j00, j02, j11, j12;
  float T0[3], T1[3];
  float a_raw, b, c_raw, a, c;
};

// EWA projection of the 3-D covariance (/root/reference/LoG/cuda/compute_radius_kernel.cu:63-105),
// with the low-pass selectable: fork max(.,0.3) (:102-103) / upstream +0.3 (LoG/model/geometry.py:87-88) / none.
// (V: const float*, or the view's matrix through lr_uniform())
// (lr_ewa_t: everything behind the view-space centre t -- the batched projection computes t early, together with the
// other values that need the Gaussian's raw inputs, so that the next Gaussian's inputs can be requested into the same
// registers; lr_ewa = t, then lr_ewa_t: one op sequence)
template <typename VP>
LR_DEV void lr_ewa_t(const float Sg[6], VP V, float fx, float fy, float tanfovx, float tanfovy, int filter_mode,
                     LrEwa& e) {
  float tz = e.t[2];
  float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
  float txtz = e.t[0] / tz, tytz = e.t[1] / tz;
  e.cx = (txtz < -limx) || (txtz > limx);
  e.cy = (tytz < -limy) || (tytz > limy);
  e.ux = fminf(limx, fmaxf(-limx, txtz));
  e.uy = fminf(limy, fmaxf(-limy, tytz));
  e.txc = e.ux * tz;
  e.tyc = e.uy * tz;
  float tz2 = tz * tz;
  e.j00 = fx / tz; e.j02 = -(fx * e.txc) / tz2;
  e.j11 = fy / tz; e.j12 = -(fy * e.tyc) / tz2;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    e.T0[j] = lr_fma(e.j00, V[4 * j + 0], e.j02 * V[4 * j + 2]);
    e.T1[j] = lr_fma(e.j11, V[4 * j + 1], e.j12 * V[4 * j + 2]);
  }
  float w0[3], w1[3];
  w0[0] = lr_dot3(Sg[0], Sg[1], Sg[2], e.T0[0], e.T0[1], e.T0[2]);
  w0[1] = lr_dot3(Sg[1], Sg[3], Sg[4], e.T0[0], e.T0[1], e.T0[2]);
  w0[2] = lr_dot3(Sg[2], Sg[4], Sg[5], e.T0[0], e.T0[1], e.T0[2]);
  w1[0] = lr_dot3(Sg[0], Sg[1], Sg[2], e.T1[0], e.T1[1], e.T1[2]);
  w1[1] = lr_dot3(Sg[1], Sg[3], Sg[4], e.T1[0], e.T1[1], e.T1[2]);
  w1[2] = lr_dot3(Sg[2], Sg[4], Sg[5], e.T1[0], e.T1[1], e.T1[2]);
  e.a_raw = lr_dot3(e.T0[0], e.T0[1], e.T0[2], w0[0], w0[1], w0[2]);
  e.b = lr_dot3(e.T0[0], e.T0[1], e.T0[2], w1[0], w1[1], w1[2]);
  e.c_raw = lr_dot3(e.T1[0], e.T1[1], e.T1[2], w1[0], w1[1], w1[2]);
  if (filter_mode == LOGRAST_FILTER_DILATE) { e.a = e.a_raw + 0.3f; e.c = e.c_raw + 0.3f; }
  else if (filter_mode == LOGRAST_FILTER_CLAMP) { e.a = fmaxf(e.a_raw, 0.3f); e.c = fmaxf(e.c_raw, 0.3f); }
  else { e.a = e.a_raw; e.c = e.c_raw; }
}
template <typename VP>
LR_DEV void lr_ewa(const float p[3], const float Sg[6], VP V, float fx, float fy,
                   float tanfovx, float tanfovy, int filter_mode, LrEwa& e) {
  e.t[0] = lr_dot3p(V[0], V[4], V[8], p[0], p[1], p[2], V[12]);
  e.t[1] = lr_dot3p(V[1], V[5], V[9], p[0], p[1], p[2], V[13]);
  e.t[2] = lr_dot3p(V[2], V[6], V[10], p[0], p[1], p[2], V[14]);
  lr_ewa_t(Sg, V, fx, fy, tanfovx, tanfovy, filter_mode, e);
}

LR_DEV float lr_radius_from_cov(float a, float c, float det) {
  float mid = 0.5f * (a + c);
  float disc = fmaxf(0.1f, mid * mid - det);
  float sq = sqrtf(disc);
  float l1 = mid + sq, l2 = mid - sq;
  return 3.f * sqrtf(fmaxf(l1, l2));
}

// A0 for one Gaussian: the body of compute_radius_cuda (/root/reference/LoG/cuda/compute_radius_kernel.cu:107-156)
// -- float radius (no ceil), |ndc| > 1.3 cull only, fork low-pass max(., 0.3), det == 0 -> 0.  Shared by
// lr_radius_kernel (project.hip) and the level-of-detail traversal (lod.hip).
LR_DEV float lr_radius_one(const float p[3], const float s[3], const float q[4], const float* __restrict__ proj,
                           const float* __restrict__ view, float fx, float fy, float tanfovx, float tanfovy) {
  float hx = lr_dot3p(proj[0], proj[4], proj[8], p[0], p[1], p[2], proj[12]);
  float hy = lr_dot3p(proj[1], proj[5], proj[9], p[0], p[1], p[2], proj[13]);
  float hw = lr_dot3p(proj[3], proj[7], proj[11], p[0], p[1], p[2], proj[15]);
  float pw = 1.0f / (hw + 0.0000001f);
  float nx = hx * pw, ny = hy * pw;
  float out = 0.f;
  if (!(nx < -1.3f || nx > 1.3f || ny < -1.3f || ny > 1.3f)) {
    float R[9], Sg[6];
    lr_cov3d(s, q, R, Sg);
    LrEwa e;
    lr_ewa(p, Sg, view, fx, fy, tanfovx, tanfovy, LOGRAST_FILTER_CLAMP, e);
    float det = e.a * e.c - e.b * e.b;
    if (det != 0.0f) out = lr_radius_from_cov(e.a, e.c, det);
  }
  return out;
}

// exp(x) for either sign: lr_exp's op sequence with the exponent also clamped from above (lr_exp itself stays as
// it is: the blend kernels only call it with x <= 0).  Shared with oracle/lograst_oracle.c:ora_exp_any.
LR_DEV float lr_exp_any(float x) {
  float t = x * 1.44269504088896341f;
  t = fminf(fmaxf(t, -125.0f), 125.0f);
  float n = rintf(t);
  float f = t - n;
  float p = 0x1.5c08e4p-10f;
  p = lr_fma(p, f, 0x1.3d0c52p-7f);
  p = lr_fma(p, f, 0x1.c6b6e4p-5f);
  p = lr_fma(p, f, 0x1.ebf918p-3f);
  p = lr_fma(p, f, 0x1.62e428p-1f);
  p = lr_fma(p, f, 0x1.000002p+0f);
  int ni = (int)n;
  return __uint_as_float(__float_as_uint(p) + ((uint32_t)ni << 23));
}

// torch.nn.functional.normalize on one quaternion (LoG/model/activation.py:17): q / max(|q|, 1e-12), as a fixed
// op sequence (oracle: ora_normalize4).
LR_DEV void lr_normalize4(float q[4]) {
  const float n2 = lr_fma(q[0], q[0], lr_fma(q[1], q[1], lr_fma(q[2], q[2], q[3] * q[3])));
  const float d = fmaxf(sqrtf(n2), 1e-12f);
  q[0] = q[0] / d; q[1] = q[1] / d; q[2] = q[2] / d; q[3] = q[3] / d;
}

// ---- alpha-support test against a pixel box, prepared once per Gaussian ------------------------------
// Can the Gaussian reach alpha >= 1/255 anywhere in the pixel box [x0,x1]x[y0,y1]?  Same conservative rule as
// blend.hip:lr_support_hits (level set 0.5 d^T Q d <= 1.01 ln(255 opacity) + 0.01, "yes" whenever the fp32
// evaluation of `power` could exceed that margin), split so that the binning kernels can test every tile of a
// rect against one prepared Gaussian.  Used by project + fill, which must agree bit for bit (same inputs: the
// stored record; same code: this function).
struct LrSupport {
  float mx, my, A, B, C, tau, ex, ey, iA, iC;
  int mode;  // 0 never visible (opacity below the floor), 1 cannot cull safely, 2 test the box
};
LR_DEV LrSupport lr_support_prepare(float mx, float my, float A, float B, float C, float op) {
  // v_rcp_f32 / v_sqrt_f32 (1 ulp) instead of the IEEE sequences: the test only has to be conservative and
  // reproducible (project and fill run this same code on the same record), and its margins are 1 % wide.
  // Straight-line: every field is computed for every lane, the mode is selected at the end (the projection runs this
  // once per Gaussian; branches around it cost more scalar bookkeeping than the ~25 instructions they skip).
  LrSupport s;
  s.mx = mx; s.my = my; s.A = A; s.B = B; s.C = C;
  s.tau = lr_fma(__logf(255.f * op), 1.01f, 0.01f);
  const float det = A * C - B * B;
  const float inv = __builtin_amdgcn_rcpf(det);
  const float ex2 = 2.f * s.tau * C * inv, ey2 = 2.f * s.tau * A * inv;  // squared half extents
  const float mag = (fabsf(A) + fabsf(B) + fabsf(C)) * (ex2 + ey2);       // bound on the terms of `power` in the box
  const bool safe = (det > 0.f) && (ex2 >= 0.f) && (ey2 >= 0.f) && (mag * 1.0e-6f < 0.005f * s.tau) && (mag < 1.0e30f);
  s.ex = __builtin_amdgcn_sqrtf(ex2) * 1.0001f + 0.01f; s.ey = __builtin_amdgcn_sqrtf(ey2) * 1.0001f + 0.01f;
  s.iA = __builtin_amdgcn_rcpf(A); s.iC = __builtin_amdgcn_rcpf(C);
  // opacity below the floor: never visible (0); NaN opacity or an ill-conditioned conic: cannot cull safely (1)
  s.mode = (op < 1.0f / 512.0f) ? 0 : ((op >= 1.0f / 512.0f && safe) ? 2 : 1);
  return s;
}
// The minimum of the quadratic over the box lies -- unless the centre is inside -- on an edge that FACES the centre: a
// point of an averted edge sees the centre through the box, and the (convex) quadratic falls along that segment.  So one
// vertical and one horizontal edge are evaluated (the facing one; either when the centre lies between the two), each
// at the minimum of the quadratic along it (closed form, clamped to the edge).
LR_DEV bool lr_support_box(const LrSupport& s, float x0, float x1, float y0, float y1) {
  if (s.mode != 2) return s.mode != 0;
  if (!((s.mx + s.ex >= x0) && (s.mx - s.ex <= x1) && (s.my + s.ey >= y0) && (s.my - s.ey <= y1))) return false;
  const float dx0 = (x0 - 0.01f) - s.mx, dx1 = (x1 + 0.01f) - s.mx;
  const float dy0 = (y0 - 0.01f) - s.my, dy1 = (y1 + 0.01f) - s.my;
  if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return true;
  const float dx = dx0 > 0.f ? dx0 : dx1;
  const float dy = fminf(dy1, fmaxf(dy0, -s.B * dx * s.iC));
  const float bv = 0.5f * (s.A * dx * dx + s.C * dy * dy) + s.B * dx * dy;
  const float ey_ = dy0 > 0.f ? dy0 : dy1;
  const float ex_ = fminf(dx1, fmaxf(dx0, -s.B * ey_ * s.iA));
  const float bh = 0.5f * (s.A * ex_ * ex_ + s.C * ey_ * ey_) + s.B * ex_ * ey_;
  return !(fminf(bv, bh) > s.tau);
}
LR_DEV bool lr_support_tile(const LrSupport& s, int tx, int ty) {
  const float x0 = (float)(tx * LR_TILE), y0 = (float)(ty * LR_TILE);
  return lr_support_box(s, x0, x0 + (float)(LR_TILE - 1), y0, y0 + (float)(LR_TILE - 1));
}
// Two boxes at once ([X0, X1] x [Y0, Y1] in pixels, one box per half): element for element the op sequence of
// lr_support_box, the multiplies and adds issued as v_pk_*_f32 -- the same decisions as two lr_support_box calls.
LR_DEV void lr_support_box2(const LrSupport& s, lr_f2 X0, lr_f2 X1, lr_f2 Y0, lr_f2 Y1, bool& keep0, bool& keep1) {
  const float xlo = s.mx + s.ex, xhi = s.mx - s.ex, ylo = s.my + s.ey, yhi = s.my - s.ey;
  const bool bb0 = (xlo >= X0.x) && (xhi <= X1.x) && (ylo >= Y0.x) && (yhi <= Y1.x);
  const bool bb1 = (xlo >= X0.y) && (xhi <= X1.y) && (ylo >= Y0.y) && (yhi <= Y1.y);
  const lr_f2 dx0 = (X0 - 0.01f) - s.mx, dx1 = (X1 + 0.01f) - s.mx;
  const lr_f2 dy0 = (Y0 - 0.01f) - s.my, dy1 = (Y1 + 0.01f) - s.my;
  const bool in0 = dx0.x <= 0.f && dx1.x >= 0.f && dy0.x <= 0.f && dy1.x >= 0.f;
  const bool in1 = dx0.y <= 0.f && dx1.y >= 0.f && dy0.y <= 0.f && dy1.y >= 0.f;
  const lr_f2 dx = lr_f2{dx0.x > 0.f ? dx0.x : dx1.x, dx0.y > 0.f ? dx0.y : dx1.y};
  const lr_f2 ty_ = -s.B * dx * s.iC;
  const lr_f2 dy = lr_f2{fminf(dy1.x, fmaxf(dy0.x, ty_.x)), fminf(dy1.y, fmaxf(dy0.y, ty_.y))};
  const lr_f2 bv = 0.5f * (s.A * dx * dx + s.C * dy * dy) + s.B * dx * dy;
  const lr_f2 ey_ = lr_f2{dy0.x > 0.f ? dy0.x : dy1.x, dy0.y > 0.f ? dy0.y : dy1.y};
  const lr_f2 tx_ = -s.B * ey_ * s.iA;
  const lr_f2 ex_ = lr_f2{fminf(dx1.x, fmaxf(dx0.x, tx_.x)), fminf(dx1.y, fmaxf(dx0.y, tx_.y))};
  const lr_f2 bh = 0.5f * (s.A * ex_ * ex_ + s.C * ey_ * ey_) + s.B * ex_ * ey_;
  const bool q0 = !(fminf(bv.x, bh.x) > s.tau), q1 = !(fminf(bv.y, bh.y) > s.tau);
  const bool any = s.mode != 0, test = s.mode == 2;
  keep0 = test ? (bb0 && (in0 || q0)) : any;
  keep1 = test ? (bb1 && (in1 || q1)) : any;
}
// Two tiles at once (tile origins X0, Y0 in pixels, one tile per half).
LR_DEV void lr_support_tile2(const LrSupport& s, lr_f2 X0, lr_f2 Y0, bool& keep0, bool& keep1) {
  lr_support_box2(s, X0, X0 + (float)(LR_TILE - 1), Y0, Y0 + (float)(LR_TILE - 1), keep0, keep1);
}

// Every kernel behind the fill checks this first: nothing is sorted or composited when the caller's buffers were too
// small (capacity) or its longest-list hint too low (the fill kernel raises LR_HDR_OVERFLOW for both).
LR_DEV bool lr_bail(const uint32_t* __restrict__ state, uint32_t capacity) {
  return state[LR_HDR_NUM] > capacity || state[LR_HDR_OVERFLOW] != 0u;
}

// ---- wave64 helpers ------------------------------------------------------------------------------
LR_DEV float lr_readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
LR_DEV int lr_readlane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// DPP wave64 max; the result ends up in lane 63 (row_shr within 16-lane rows, then row broadcasts).
// Done on the raw bit patterns of NON-NEGATIVE floats (unsigned order == float order): v_max_u32 needs no NaN
// canonicalisation, so every step is one DPP-fused instruction.
template <int CTRL, int ROW_MASK, int BANK_MASK>
LR_DEV uint32_t lr_dpp_umax(uint32_t v) {
  uint32_t moved = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, true);  // 0 = umax identity
  return max(v, moved);
}
LR_DEV uint32_t lr_wave_umax_to63(uint32_t v) {
  v = lr_dpp_umax<0x111, 0xf, 0xf>(v);
  v = lr_dpp_umax<0x112, 0xf, 0xf>(v);
  v = lr_dpp_umax<0x114, 0xf, 0xe>(v);
  v = lr_dpp_umax<0x118, 0xf, 0xc>(v);
  v = lr_dpp_umax<0x142, 0xa, 0xf>(v);
  v = lr_dpp_umax<0x143, 0xc, 0xf>(v);
  return v;
}

// ---- rects of LR_RANKED_TILES + 1 ... LR_COOP_TILES tiles, expanded by the wave --------------------------------------
// Projection (counting) and fill (placing) used to walk such a rect per lane: nt iterations of a ~70-instruction support
// test + atomic with the few lanes that hold one active -- every wave that meets ONE 16-tile rect runs 16 iterations at
// 1/64 lane occupancy.  On check_gui's uniform draws (the headline) these rects do not exist; on log-normal scales (a
// trained model: 3-4 % of the Gaussians, two per wave) they doubled the fill (30 M: 307 -> 649 us).  Here the wave expands
// them together, FOUR rects per pass: lane l serves tile (l & 15) of the (4 * pass + (l >> 4))-th such rect, whose rect and
// prepared support come over from the owning lane with ds_bpermute.  Same support test on the same values, so the same
// tiles; which lane counts / places an instance does not matter (the lists are sorted afterwards).
// Call from wave-uniform control flow with all 64 lanes active (ds_bpermute reads nothing from an inactive lane).
// f(t, tile_y, tile_x, keep, pay0, pay1, pay2) runs in the serving lane for EVERY tile t < nt of the rect: keep = the tile
// passes the support test (TEST = false: no test, keep = true -- the fill of rects the projection ranked: their ranks say
// which tiles were kept); pay0..2 are the owning lane's payload words (its rank row; the fill: its key).
LR_DEV int lr_bperm_i(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
LR_DEV float lr_bperm_f(float v, int src) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v))); }
template <bool TEST, typename F>
LR_DEV void lr_mid_rects(bool mid, int x0, int y0, int w, int nt, const LrSupport& sup, int pay0, int pay1, int pay2, F&& f) {
  uint64_t m = __ballot(mid);
  const int lane = (int)threadIdx.x & 63, sub = lane >> 4, t = lane & 15;
  while (m) {
    int s[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { s[q] = m ? (int)__builtin_ctzll(m) : -1; m &= m - 1ull; }   // (0 stays 0)
    const int src = sub == 0 ? s[0] : (sub == 1 ? s[1] : (sub == 2 ? s[2] : s[3]));
    const int sl = src < 0 ? lane : src;
    const int bx0 = lr_bperm_i(x0, sl), by0 = lr_bperm_i(y0, sl), bw = lr_bperm_i(w, sl), bn = lr_bperm_i(nt, sl);
    LrSupport bs = sup;
    if (TEST) {
      bs.mx = lr_bperm_f(sup.mx, sl); bs.my = lr_bperm_f(sup.my, sl);
      bs.A = lr_bperm_f(sup.A, sl); bs.B = lr_bperm_f(sup.B, sl); bs.C = lr_bperm_f(sup.C, sl);
      bs.tau = lr_bperm_f(sup.tau, sl); bs.ex = lr_bperm_f(sup.ex, sl); bs.ey = lr_bperm_f(sup.ey, sl);
      bs.iA = lr_bperm_f(sup.iA, sl); bs.iC = lr_bperm_f(sup.iC, sl);
      bs.mode = lr_bperm_i(sup.mode, sl);
    }
    const int q0 = lr_bperm_i(pay0, sl), q1 = lr_bperm_i(pay1, sl), q2 = lr_bperm_i(pay2, sl);
    const bool have = src >= 0 && t < bn;
    const int bwc = bw > 0 ? bw : 1;
    const int ty = t / bwc, tx = t - ty * bwc;
    if (have) f(t, by0 + ty, bx0 + tx, !TEST || lr_support_tile(bs, bx0 + tx, by0 + ty), q0, q1, q2);
  }
}

// ---- ranks of the 5..16-tile rects (round 5) ---------------------------------------------------------------------------
// Counting such a rect cost the fill one returning memory-side atomic per tile instance (the per-tile cursors: ~20 G line
// operations per second chip-wide; a trained model's 3-4 % of such Gaussians made them 8 M per 30 M-Gaussian view).  The
// batched projection now RANKS them like the rects of up to four tiles -- the serving lane's LDS atomic on the (batch,
// tile) counter returns the instance's rank inside the batch -- and leaves the ranks of a rect in a 32-byte row of 16-bit
// words (0xffff = tile dropped by the support cull) that the fill reads back: no cursor atomic, no second support test,
// no record fetch.  Rows live behind the records | fill records | band indices in `geom`: batch b owns rows
// [b * lr_mid_cap(B), ...) -- one row per four Gaussians of the batch; a batch with more such rects counts the rest the old way.
#define LR_MID_ROW 16
// lr_mid_rects serves a rect with 16 lanes (t = lane & 15, four rects per pass) and a rank row holds 16 ranks: a rect handed to
// either must not have more tiles than that (round-5 advisory: nothing enforced it).
static_assert(LR_COOP_TILES <= 16 && LR_MID_ROW == 16, "lr_mid_rects / rank rows serve at most 16 tiles per rect");
__host__ __device__ inline uint32_t lr_mid_cap(uint32_t B) { return B >> 2; }   // (C2: 13 % of the Gaussians hold such a rect)
__host__ __device__ inline size_t lr_midrank_off_bytes(size_t n) {
  return ((sizeof(float) * LOGRAST_REC_FLOATS + 16 + 4) * n + 63) & ~(size_t)63;
}
// rows of all batches: batches * (B / 4) * 32 B = 8 B per Gaussian of the padded batches, and batches * B < n + 32768
__host__ __device__ inline size_t lr_midrank_bytes(size_t n) { return 8 * n + 8 * 32768; }

// ---- N4 kernel arguments (counter.hip; filled in by api.hip) ------------------------------------------
struct CounterArgs {
  const int64_t* visible_index;
  const float* grad;       // dL/dmeans2D [nv, 3]
  const int32_t* radii;
  const float* weight;     // point_weight [nv]
  const int32_t* point_id;
  const int64_t* point_count;
  float* weights_max; float* weights_sum; float* grad_sum;
  int16_t* radii_max; int16_t* visible_count;
  int32_t* radii_max_max; int32_t* area_sum; int32_t* create_steps;
  uint8_t* flag_vis;
  int32_t nv, k, num_points;
};

#define ADAM_MAX_KEYS 8
struct AdamKey {
  float* model;        // [num_points, width]: rows `index` are rewritten
  const float* param;  // [m, width]: the gathered parameter the step starts from (params[key].data)
  const float* grad;   // [m, width]
  float* exp_avg; float* exp_avg_sq; float* max_exp_avg_sq;  // [num_points, width]; max_exp_avg_sq may be NULL
  int32_t width;
  float neg_step_size;
};
struct AdamArgs {
  AdamKey key[ADAM_MAX_KEYS];
  const int64_t* index;
  const uint8_t* flag_vis;
  int32_t m, num_points;
  float beta1, beta2, omb1, omb2, bc2_sqrt, eps;
};

// ---- fused get_all (sh.hip: ga_fwd_kernel / ga_bwd_kernel) ---------------------------------------------
struct GatherArgs {
  const int64_t* index;
  const float *xyz, *scaling, *opacity, *rotation, *colors, *shs;   // model buffers [num_points, ...]
  const float* campos;
  float *r_xyz, *r_scaling, *r_opacity, *r_rotation, *r_colors, *r_shs;  // gathered raw rows [n, ...]
  float *a_scaling, *a_opacity, *a_rotation, *a_colors;                  // activated [n, ...]
  int n, num_points, K, deg;
};

struct ActBwdArgs {
  const float *r_xyz, *r_scaling, *r_opacity, *r_rotation;    // the gathered raw rows (the Parameters' data)
  const float* campos;
  const float *g_a_scaling, *g_a_opacity, *g_a_rotation, *g_a_colors;   // dL/d(activated), rows [0, n) are used
  float *g_scaling, *g_opacity, *g_rotation, *g_colors, *g_shs;           // dL/d(raw) [n, ...]; g_shs may be NULL
  int n, K, deg;
};

// ---- host-side launch bookkeeping (api.hip) ------------------------------------------------------
enum LrKernelSlot {
  LRK_RADIUS = 0, LRK_PROJECT, LRK_SCAN, LRK_FILL, LRK_SORT_SMALL, LRK_SORT_LARGE, LRK_SORT_HUGE,
  LRK_BLEND_FWD, LRK_BLEND_BWD, LRK_PROJECT_BWD, LRK_MISC, LRK_LOD, LRK_COUNTER, LRK_ADAM, LRK_HIST, LRK_GATHER, LRK_GATHER_BWD, LRK_RESERVED,
  LRK_REBASE, LRK_LAZY_TAIL   // LRK_LAZY_TAIL: the second sort + compositing pair of lazily ordered lists (normally idle)
};
// ---- experiment switches ---------------------------------------------------------------------------------------
// Timing ablations (kernels that SKIP part of their work) and alternative algorithms kept for A/B measurements exist only
// in builds made with -DLR_EXPERIMENTS (`python -m log_amd.build <variant> -DLR_EXPERIMENTS`, loaded through LOGRAST_LIB by
// the scripts under tools/); the product library contains none of them: LR_EXPERIMENT_INT is its default, no kernel
// takes an `ablate` argument, LR_ABLATED() is `false`.
#ifdef LR_EXPERIMENTS
#define LR_EXPERIMENT_INT(name, dflt) lr_env_int(name, dflt)
#define LR_ABLATE_PARAM , int ablate
#define LR_ABLATE_PASS(x) , x
#define LR_ABLATED(bits) ((ablate & (bits)) != 0)
#else
#define LR_EXPERIMENT_INT(name, dflt) (dflt)
#define LR_ABLATE_PARAM
#define LR_ABLATE_PASS(x)
#define LR_ABLATED(bits) false
#endif
void lr_prof_begin(int slot, hipStream_t s);
void lr_prof_end(int slot, hipStream_t s);
int lr_env_int(const char* name, int dflt);
// Performance knobs (api.hip): value = lograst_set_knob override, else the environment variable of that name, else the
// default.  Read at every launch through a per-call-site cache that is refreshed when any knob changes, so that
// log_amd.tune() can move them at run time.  None of them changes a result (tests/test_gpu_knobs.py sweeps them and
// compares bit for bit).
// The cache is one 64-bit atomic (generation << 32 | value): a reader sees a generation together with the value that
// was looked up FOR it -- the generation is read once, before the look-up, so a lograst_set_knob on another thread in
// between only makes the next launch look again (round-3 advisory: two plain fields, generation re-read after the look-up).
extern unsigned lr_knob_generation();
int lr_knob_lookup(const char* name, int dflt);
#define LR_KNOB(var, name, dflt)                                                                       \
  static std::atomic<uint64_t> var##_cache{~0ull};                                                     \
  int var##_value;                                                                                     \
  {                                                                                                    \
    const unsigned var##_gen = lr_knob_generation();                                                   \
    const uint64_t var##_c = var##_cache.load(std::memory_order_acquire);                              \
    if ((unsigned)(var##_c >> 32) == var##_gen) {                                                      \
      var##_value = (int)(uint32_t)var##_c;                                                            \
    } else {                                                                                           \
      var##_value = lr_knob_lookup(name, dflt);                                                        \
      var##_cache.store(((uint64_t)var##_gen << 32) | (uint32_t)var##_value, std::memory_order_release); \
    }                                                                                                  \
  }                                                                                                    \
  const int var = var##_value

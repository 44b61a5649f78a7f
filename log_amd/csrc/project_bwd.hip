// project_bwd.hip -- A6b: per-Gaussian chain rule from (dL/d ndc-mean, dL/d conic) back to
// means3D / scales / rotations.  Streaming, one thread per Gaussian; everything (cov3D, EWA Jacobian)
// is recomputed from the original inputs instead of being stored by the forward (saves 24+ B/Gaussian
// of HBM traffic each way).  Conventions: SURVEY.md Appendix B; the alpha cap and the 0.1 floor in the
// radius formula carry no gradient; the fork's max(.,0.3) low-pass has the standard sub-gradient.
#include "common.hpp"

// TOUCHED (lograst_backward with a point_weight array): a Gaussian whose forward blend weight stayed 0 contributed
// to no pixel, so the reverse walk added nothing to its dL/dmean2D and dL/dconic -- both are exactly zero and so is
// everything the chain rule would compute from them.  Such rows are skipped without reading their 56 input bytes or
// their accumulators (whose conic part the forward then does not zero-fill: the compositing kernel clears the rows it meets) and without the 80-byte read-modify-
// write of the running sums: in an opaque scene most Gaussians are hidden (30 M random Gaussians at opacity 0.999:
// the kernel went from 0.83 ms, at the copy rate, to the touched rows' share).
// The chain rule of ONE Gaussian the forward composited (gm, gs, gq, and dL/dcov3D written / added in place when COV).
template <bool ACCUMULATE, bool COV>
LR_DEV void lr_project_bwd_row(const LrView& v, int i, const float* __restrict__ means, const float* __restrict__ scales,
                               const float* __restrict__ rots, float gnx, float gny, float gA, float gB, float gC,
                               float gm[3], float gs[3], float gq[4]) {
  const float* __restrict__ V = v.view;
  const float* __restrict__ Pm = v.proj;
  float p[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
  float s[3] = {0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  float R[9], Sg[6];
  if (COV) {   // cov3D_precomp (the view carries n x 6 covariances): the chain stops at dL/dSigma
#pragma unroll
    for (int k = 0; k < 6; k++) Sg[k] = v.cov3d[6 * (size_t)i + k];
  } else {
    s[0] = scales[3 * i] * v.scale_modifier; s[1] = scales[3 * i + 1] * v.scale_modifier;
    s[2] = scales[3 * i + 2] * v.scale_modifier;
    const float4 q4 = reinterpret_cast<const float4*>(rots)[i];
    q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
    lr_cov3d(s, q, R, Sg);
  }
  LrEwa e;
  lr_ewa(p, Sg, V, v.fx, v.fy, v.tanfovx, v.tanfovy, v.filter_mode, e);
  const float a = e.a, b = e.b, c = e.c;
  const float det = a * c - b * b;
  const float di2 = 1.f / (det * det);
  // conic = (c, -b, a) / det
  float ga = di2 * (-c * c * gA + b * c * gB - b * b * gC);
  float gb = di2 * (2.f * b * c * gA - (det + 2.f * b * b) * gB + 2.f * a * b * gC);
  float gc = di2 * (-b * b * gA + a * b * gB - a * a * gC);
  if (v.filter_mode == LOGRAST_FILTER_CLAMP) {
    if (!(e.a_raw >= 0.3f)) ga = 0.f;
    if (!(e.c_raw >= 0.3f)) gc = 0.f;
  }
  const float hb = 0.5f * gb;
  // dL/dSigma = T^T G2 T
  float gS[9];
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int k = 0; k < 3; k++)
      gS[3 * j + k] = ga * e.T0[j] * e.T0[k] + hb * (e.T0[j] * e.T1[k] + e.T1[j] * e.T0[k]) + gc * e.T1[j] * e.T1[k];
  // dL/dT = 2 G2 T Sigma
  const float S3[9] = {Sg[0], Sg[1], Sg[2], Sg[1], Sg[3], Sg[4], Sg[2], Sg[4], Sg[5]};
  float gT0[3], gT1[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float ts0 = e.T0[0] * S3[k] + e.T0[1] * S3[3 + k] + e.T0[2] * S3[6 + k];
    float ts1 = e.T1[0] * S3[k] + e.T1[1] * S3[3 + k] + e.T1[2] * S3[6 + k];
    gT0[k] = 2.f * (ga * ts0 + hb * ts1);
    gT1[k] = 2.f * (hb * ts0 + gc * ts1);
  }
  float gj00 = 0.f, gj02 = 0.f, gj11 = 0.f, gj12 = 0.f;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    gj00 += gT0[j] * V[4 * j + 0]; gj02 += gT0[j] * V[4 * j + 2];
    gj11 += gT1[j] * V[4 * j + 1]; gj12 += gT1[j] * V[4 * j + 2];
  }
  const float tz = e.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
  const float g_txc = -v.fx / tz2 * gj02, g_tyc = -v.fy / tz2 * gj12;
  float g_tz = -v.fx / tz2 * gj00 - v.fy / tz2 * gj11 + 2.f * v.fx * e.txc / tz3 * gj02 +
               2.f * v.fy * e.tyc / tz3 * gj12;
  float g_tx, g_ty;
  if (e.cx) { g_tx = 0.f; g_tz += e.ux * g_txc; } else { g_tx = g_txc; }
  if (e.cy) { g_ty = 0.f; g_tz += e.uy * g_tyc; } else { g_ty = g_tyc; }
  float m0 = V[0] * g_tx + V[1] * g_ty + V[2] * g_tz;
  float m1 = V[4] * g_tx + V[5] * g_ty + V[6] * g_tz;
  float m2 = V[8] * g_tx + V[9] * g_ty + V[10] * g_tz;
  // centre path: ndc = h.xy / (h.w + eps)
  const float hx = lr_dot3p(Pm[0], Pm[4], Pm[8], p[0], p[1], p[2], Pm[12]);
  const float hy = lr_dot3p(Pm[1], Pm[5], Pm[9], p[0], p[1], p[2], Pm[13]);
  const float hw = lr_dot3p(Pm[3], Pm[7], Pm[11], p[0], p[1], p[2], Pm[15]);
  const float pw = 1.0f / (hw + 0.0000001f);
  const float ghx = gnx * pw, ghy = gny * pw, ghw = -(gnx * hx + gny * hy) * pw * pw;
  m0 += Pm[0] * ghx + Pm[1] * ghy + Pm[3] * ghw;
  m1 += Pm[4] * ghx + Pm[5] * ghy + Pm[7] * ghw;
  m2 += Pm[8] * ghx + Pm[9] * ghy + Pm[11] * ghw;
  gm[0] = m0; gm[1] = m1; gm[2] = m2;
  if (COV) {
    // the six stored entries; an off-diagonal one stands for both symmetric positions of Sigma (the upstream backward's
    // "off-diagonal elements appear twice" rule)
    const float g6[6] = {gS[0], 2.f * gS[1], 2.f * gS[2], gS[4], 2.f * gS[5], gS[8]};
    float* __restrict__ o = v.g_cov3d + 6 * (size_t)i;
#pragma unroll
    for (int k = 0; k < 6; k++) o[k] = ACCUMULATE ? o[k] + g6[k] : g6[k];
  } else {
  // Sigma = M M^T, M_ik = R_ik s_k
  float M[9], gM[9], gR[9];
#pragma unroll
  for (int ii = 0; ii < 3; ii++)
#pragma unroll
    for (int k = 0; k < 3; k++) M[3 * ii + k] = R[3 * ii + k] * s[k];
#pragma unroll
  for (int ii = 0; ii < 3; ii++)
#pragma unroll
    for (int k = 0; k < 3; k++)
      gM[3 * ii + k] = 2.f * (gS[3 * ii + 0] * M[0 + k] + gS[3 * ii + 1] * M[3 + k] + gS[3 * ii + 2] * M[6 + k]);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    gs[k] = v.scale_modifier * (gM[k] * R[k] + gM[3 + k] * R[3 + k] + gM[6 + k] * R[6 + k]);
#pragma unroll
    for (int ii = 0; ii < 3; ii++) gR[3 * ii + k] = gM[3 * ii + k] * s[k];
  }
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  gq[0] = 2.f * (-z * gR[1] + y * gR[2] + z * gR[3] - x * gR[5] - y * gR[6] + x * gR[7]);
  gq[1] = 2.f * (y * gR[1] + z * gR[2] + y * gR[3] - 2.f * x * gR[4] - r * gR[5] + z * gR[6] + r * gR[7] - 2.f * x * gR[8]);
  gq[2] = 2.f * (-2.f * y * gR[0] + x * gR[1] + r * gR[2] + x * gR[3] + z * gR[5] - r * gR[6] + z * gR[7] - 2.f * y * gR[8]);
  gq[3] = 2.f * (-2.f * z * gR[0] - r * gR[1] + x * gR[2] + r * gR[3] - 2.f * z * gR[4] + y * gR[5] + x * gR[6] + y * gR[7]);
  }
}

// The chain rule of ONE live Gaussian i: reads its accumulator row (AOS) or its (dL/dmean2D, dL/dconic) pair, hands out
// the per-view outputs and writes / adds the gradients (see lr_project_bwd_kernel for the template flags).
template <bool ACCUMULATE, bool COV, bool AOS, bool SINKROWS>
// (No __restrict__ here: the kernels' own parameters carry it.  Repeated on this inlined function it gave the scheduler
// licence to hoist every load of the row above the chain rule -- 136 -> 170 VGPRs, three waves per SIMD -> two, the
// 30 M-row launch 402 -> 477 us.)
LR_DEV void lr_pbwd_live_row(const LrView& v, int i, const float* means, const float* scales, const float* rots,
                             const float* g_mean2d, const float* g_conic, const float4* rows, float* o_mean2d,
                             float* o_opac, float* o_col, float* g_means3d, float* g_scales, float* g_rots) {
    float gm[3], gs[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
    float gnx, gny, gA, gB, gC;
    // running sums (ACCUMULATE): every old value is requested up front, next to the row's inputs -- the row is a chain of
    // ~20 scattered accesses, and the read-modify-writes behind the chain rule's ~600 instructions were fully exposed
    float old_m[3] = {0.f, 0.f, 0.f}, old_s[3] = {0.f, 0.f, 0.f}, old_c[3] = {0.f, 0.f, 0.f}, old_o = 0.f;
    float4 old_q = {0.f, 0.f, 0.f, 0.f};
    float4 sr0 = {0.f, 0.f, 0.f, 0.f}, sr1 = sr0, sr2 = sr0, sr3 = sr0;
    float4* const srow = SINKROWS ? reinterpret_cast<float4*>(g_means3d) + 4 * (size_t)i : nullptr;
    if (SINKROWS) {
      sr0 = srow[0]; sr1 = srow[1]; sr2 = srow[2]; sr3 = srow[3];
    } else if (ACCUMULATE) {
#pragma unroll
      for (int k = 0; k < 3; k++) old_m[k] = g_means3d[3 * (size_t)i + k];
      if (!COV) {
#pragma unroll
        for (int k = 0; k < 3; k++) old_s[k] = g_scales[3 * (size_t)i + k];
        old_q = reinterpret_cast<const float4*>(g_rots)[i];
      }
      if (AOS) {
        old_o = o_opac[i];
#pragma unroll
        for (int k = 0; k < 3; k++) old_c[k] = o_col[3 * (size_t)i + k];
      }
    }
    if (AOS) {
      const float4 a0 = rows[4 * (size_t)i], a1 = rows[4 * (size_t)i + 1];
      const float cb = reinterpret_cast<const float*>(rows + 4 * (size_t)i + 2)[0];
      gnx = a0.x; gny = a0.y; gA = a0.z; gB = a0.w; gC = a1.x;
      o_mean2d[3 * (size_t)i + 0] = gnx; o_mean2d[3 * (size_t)i + 1] = gny; o_mean2d[3 * (size_t)i + 2] = 0.f;
      if (SINKROWS) {
        sr2.z += a1.y; sr2.w += a1.z; sr3.x += a1.w; sr3.y += cb;   // slots 10 opacity, 11-13 colour
      } else {
        o_opac[i] = old_o + a1.y;
        o_col[3 * (size_t)i + 0] = old_c[0] + a1.z; o_col[3 * (size_t)i + 1] = old_c[1] + a1.w; o_col[3 * (size_t)i + 2] = old_c[2] + cb;
      }
    } else {
      gnx = g_mean2d[3 * (size_t)i]; gny = g_mean2d[3 * (size_t)i + 1];
      const float4 gc4 = reinterpret_cast<const float4*>(g_conic)[i];
      gA = gc4.x; gB = gc4.y; gC = gc4.z;
    }
    lr_project_bwd_row<ACCUMULATE, COV>(v, i, means, scales, rots, gnx, gny, gA, gB, gC, gm, gs, gq);
    if (SINKROWS) {    // running sums over views, one row per Gaussian
      sr0.x += gm[0]; sr0.y += gm[1]; sr0.z += gm[2]; sr0.w += gs[0];
      sr1.x += gs[1]; sr1.y += gs[2]; sr1.z += gq[0]; sr1.w += gq[1];
      sr2.x += gq[2]; sr2.y += gq[3];
      srow[0] = sr0; srow[1] = sr1; srow[2] = sr2; srow[3] = sr3;
    } else if (ACCUMULATE) {  // running sums over views (log_amd.dist)
#pragma unroll
      for (int k = 0; k < 3; k++) g_means3d[3 * (size_t)i + k] = old_m[k] + gm[k];
      if (!COV) {
#pragma unroll
        for (int k = 0; k < 3; k++) g_scales[3 * (size_t)i + k] = old_s[k] + gs[k];
        reinterpret_cast<float4*>(g_rots)[i] = float4{old_q.x + gq[0], old_q.y + gq[1], old_q.z + gq[2], old_q.w + gq[3]};
      }
    } else {
      g_means3d[3 * (size_t)i + 0] = gm[0]; g_means3d[3 * (size_t)i + 1] = gm[1]; g_means3d[3 * (size_t)i + 2] = gm[2];
      if (!COV) {
        g_scales[3 * (size_t)i + 0] = gs[0]; g_scales[3 * (size_t)i + 1] = gs[1]; g_scales[3 * (size_t)i + 2] = gs[2];
        reinterpret_cast<float4*>(g_rots)[i] = float4{gq[0], gq[1], gq[2], gq[3]};
      }
    }
}

// One workgroup owns LR_PBWD_ROWS consecutive Gaussians.  Their live flags are read coalesced; the live rows are
// COMPACTED into an LDS list and the chain rule (~600 VALU instructions per row) then runs on full waves: with 15 % of
// the rows live (30 M Gaussians, opacity 0.999) every wave of a one-thread-per-Gaussian kernel still met a live lane and
// ran all of it at 15 % lane occupancy (VALU busy 71 % of the launch).
#ifndef LR_PBWD_ROWS
#define LR_PBWD_ROWS 1024
#endif
// AOS (lograst_backward): the reverse walk's nine sums of a Gaussian sit in ONE 64-byte row of `rows` (slots: 0-1 mean
// x y, 2-4 conic A B C, 5 opacity, 6-8 colour r g b: include/lograst.h LOGRAST_BWD_ROW_FLOATS) -- a memory-side atomic
// costs one operation per 64-byte LINE whatever the number of lanes in it (tools/micro/atomic_lines.hip: 17-21 G lines/s
// chip-wide for 1, 9 or 16 lanes per line), so the reverse walk commits a (wave, Gaussian) visit with one line operation
// instead of four.  This kernel then also hands out the API's separate outputs: dL/dmeans2D (written for every row: it
// is a per-view output), dL/dopacity and dL/dcolour (written, or added to the running sums when ACCUMULATE).
// !AOS (lograst_project_backward, the isolated chain rule): g_mean2d [n, 3] / g_conic [n, 4] in, as before.
// SINKROWS (lograst_backward with LOGRAST_BWD_ACCUMULATE_ROWS; implies ACCUMULATE and AOS, not COV): the caller's running
// sums are ONE 64-byte row per Gaussian too (`g_means3d` = [N][16]: slots 0-2 dL/dmeans3D, 3-5 dL/dscales, 6-9
// dL/drotations, 10 dL/dopacity, 11-13 dL/dcolour, 14-15 untouched) -- a live Gaussian then costs one read-modify-write of
// one line instead of five in five arrays (the live rows are scattered: at 30 M Gaussians 7 % of them, each piece of 4-16
// bytes pulling its own 64-byte line through the memory system in both directions).
// WAVES: waves per SIMD the register allocator aims for (1 = its own choice: ~130 VGPRs, three waves).  Measured, 30 M
// Gaussians, running sums / fresh gradients: default 374 / 736 us, 4 waves (128 VGPRs, no scratch) 391 / 865, 5 (spills)
// 473 / 877, 6: 518 / 918 -- more resident workgroups make the scattered rows' traffic worse, not better.  A band view
// (nearly all flag pass: a fraction of a percent of 100 M rows is live) gains from four: 738 -> 638 us -- the row-major
// sink's kernel exists in both forms and band views launch the second.
template <bool ACCUMULATE, bool TOUCHED, bool COV, bool AOS, bool SINKROWS = false, int WAVES = 1>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES)))
lr_project_bwd_kernel(LrView v, int N, const float* __restrict__ means, const float* __restrict__ scales,
                      const float* __restrict__ rots, const int* __restrict__ radii,
                      const float* __restrict__ g_mean2d, const float* __restrict__ g_conic,
                      const float4* __restrict__ rows, float* __restrict__ o_mean2d, float* __restrict__ o_opac,
                      float* __restrict__ o_col,
                      const float* __restrict__ pw, float* __restrict__ g_means3d, float* __restrict__ g_scales,
                      float* __restrict__ g_rots, int clear_inside) {
  __shared__ uint32_t live_list[LR_PBWD_ROWS];
  __shared__ uint32_t live_count;
  const int tid = threadIdx.x, base = blockIdx.x * LR_PBWD_ROWS;
  if (tid == 0) live_count = 0u;
  // the live flags of the workgroup's rows, requested before anything else and all at once: most workgroups hold no live
  // row at all, and their whole life was clear -> barrier -> radii -> (radii > 0 ?) point_weight -> barrier, one memory
  // round trip after the other (30 M Gaussians: 0.21 of the kernel's 0.43 ms with not a single live row)
  int rad_k[LR_PBWD_ROWS / 256];
  float pw_k[LR_PBWD_ROWS / 256];
#pragma unroll
  for (int k = 0; k < LR_PBWD_ROWS / 256; k++) {
    const int i = base + k * 256 + tid;
    // (TOUCHED: the forward cleared point_weight for every row and only composited Gaussians -- radii > 0 -- ever raised
    // it: the blend weight alone is the live flag, radii is not read: 0.4 of the 2.0 GB a 100 M-row band view streams)
    rad_k[k] = TOUCHED ? 1 : (i < N ? radii[i] : 0);
    pw_k[k] = (TOUCHED && i < N) ? pw[i] : 1.f;
  }
  if (AOS && clear_inside) {
    // the per-view / per-call outputs are defined for every row: the block's whole slice is cleared with full-width
    // stores first (row-by-row 12-byte stores from the flag pass below cost the 30 M view 0.2 ms), the live rows
    // overwrite theirs after the barrier.  (On large inputs lr_zero_floats_kernel streams the zeros before this kernel
    // runs: lr_launch_project_bwd.)
    const int rows_here = min(LR_PBWD_ROWS, N - base);
    typedef float lr_f4v __attribute__((ext_vector_type(4)));
    // (a scalar head up to the first 16-byte boundary, then full-width stores, then a scalar tail: the C ABI asks only
    // for 4-byte alignment of these outputs -- round-4 advisory; torch's allocations always take the head-less path)
    auto clear = [&](float* p0, int floats) {
      const int head = min(floats, (int)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(p0) & 15u)) & 15u) >> 2));
      if (tid < head) p0[tid] = 0.f;
      float* p = p0 + head;
      floats -= head;
      const int n4 = floats >> 2;
      lr_f4v* q = reinterpret_cast<lr_f4v*>(p);
      for (int t = tid; t < n4; t += 256) q[t] = lr_f4v{0.f, 0.f, 0.f, 0.f};
      for (int t = (n4 << 2) + tid; t < floats; t += 256) p[t] = 0.f;
    };
    if (rows_here > 0) {
      clear(o_mean2d + 3 * (size_t)base, 3 * rows_here);
      if (!ACCUMULATE && !SINKROWS) {
        // fresh gradients (the autograd route: what unmodified LoG gets): every output is defined for every row -- the
        // whole slice of each array with full-width stores here, instead of 12-byte pieces per dead row from the flag
        // pass below (three store instructions covering a third of every line each)
        clear(o_opac + (size_t)base, rows_here); clear(o_col + 3 * (size_t)base, 3 * rows_here);
        clear(g_means3d + 3 * (size_t)base, 3 * rows_here);
        if (COV) {
          clear(v.g_cov3d + 6 * (size_t)base, 6 * rows_here);
        } else {
          clear(g_scales + 3 * (size_t)base, 3 * rows_here);
          clear(g_rots + 4 * (size_t)base, 4 * rows_here);
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < LR_PBWD_ROWS / 256; k++) {
    const int i = base + k * 256 + tid;
    const bool live = i < N && rad_k[k] > 0 && (!TOUCHED || pw_k[k] > 0.f);
    if (!ACCUMULATE && !AOS && i < N && !live) {   // isolated chain rule: culled / untouched rows get zero gradients (AOS: cleared above)
      g_means3d[3 * (size_t)i + 0] = 0.f; g_means3d[3 * (size_t)i + 1] = 0.f; g_means3d[3 * (size_t)i + 2] = 0.f;
      if (COV) {
#pragma unroll
        for (int c = 0; c < 6; c++) v.g_cov3d[6 * (size_t)i + c] = 0.f;
      } else {
        g_scales[3 * (size_t)i + 0] = 0.f; g_scales[3 * (size_t)i + 1] = 0.f; g_scales[3 * (size_t)i + 2] = 0.f;
        reinterpret_cast<float4*>(g_rots)[i] = float4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const uint64_t m = __ballot(live);
    if (m) {
      uint32_t first = 0;
      if ((tid & 63) == 0) first = atomicAdd(&live_count, (uint32_t)__popcll(m));
      first = (uint32_t)__shfl((int)first, 0);
      if (live) live_list[first + (uint32_t)__popcll(m & ((1ull << (tid & 63)) - 1ull))] = (uint32_t)i;
    }
  }
  __syncthreads();
  const uint32_t n = live_count;
  for (uint32_t j = (uint32_t)tid; j < n; j += 256u)
    lr_pbwd_live_row<ACCUMULATE, COV, AOS, SINKROWS>(v, (int)live_list[j], means, scales, rots, g_mean2d, g_conic, rows,
                                                     o_mean2d, o_opac, o_col, g_means3d, g_scales, g_rots);
}


// ---- the chain rule over a COMPACT list of the live rows (round 5) ---------------------------------------------------------
// On large inputs whose gradients are running sums (nothing is written for a dead row) the kernel above spends half its
// time deciding that rows are dead: 29 K workgroups of a 30 M-row view, each a chain of flag load -> barrier -> (no live
// row) at three resident workgroups per CU (its chain rule needs ~130 VGPRs) -- 0.21 of its 0.40 ms; a band view of 100 M
// rows 0.5 of its 0.74.  So the decision gets a kernel of its own, shaped like a copy: `lr_pbwd_compact_kernel` streams
// point_weight (16 bytes per lane, low registers, full occupancy) and appends the indices of the rows with weight > 0 to a
// list, one slot-reserving atomic per 8192 rows; `lr_pbwd_list_kernel` then runs the chain rule over that list with a
// fixed grid (the count never travels to the host) and every lane busy.  The list needs no buffer of its own: slots 12-15
// of the reverse walk's 64-byte accumulator rows are unused (LOGRAST_BWD_ROW_FLOATS: nine sums per row), so entry j sits
// in row 1 + j / 4, slot 12 + j % 4, and the count in row 0, slot 12 (cleared by lr_zero_words_kernel before the pass).
// Same per-row code (lr_pbwd_live_row), so the same gradients bit for bit.
#define LR_PBWD_CHUNK 8192             // rows per workgroup of the compaction pass (one atomic each; 32 KB of LDS for a chunk that is all live)
LR_DEV uint32_t* lr_pbwd_list_slot(float4* rows, uint32_t j) {
  return reinterpret_cast<uint32_t*>(rows + 4 * (size_t)(1u + (j >> 2)) + 3) + (j & 3u);
}
__global__ void __launch_bounds__(1024)
lr_pbwd_compact_kernel(const float* __restrict__ pw, int N, float4* __restrict__ rows) {
  __shared__ uint32_t live[LR_PBWD_CHUNK];                    // worst case: every row of the chunk is live
  __shared__ uint32_t cnt, base_s;
  const int tid = threadIdx.x;
  const int base = blockIdx.x * LR_PBWD_CHUNK;
  if (tid == 0) cnt = 0u;
  __syncthreads();
  typedef float lr_f4v __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int k = 0; k < LR_PBWD_CHUNK / 4096; k++) {           // four rows per lane and round, all rounds requested up front
    const int i0 = base + (k * 1024 + tid) * 4;
    float w[4] = {0.f, 0.f, 0.f, 0.f};
    if (i0 + 3 < N && ((reinterpret_cast<uintptr_t>(pw) & 15u) == 0u)) {
      const lr_f4v q = __builtin_nontemporal_load(reinterpret_cast<const lr_f4v*>(pw + i0));
      w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
    } else {
#pragma unroll
      for (int u = 0; u < 4; u++) w[u] = (i0 + u < N) ? pw[i0 + u] : 0.f;
    }
    const int n_live = (w[0] > 0.f) + (w[1] > 0.f) + (w[2] > 0.f) + (w[3] > 0.f);
    // position inside the chunk's list: wave prefix over the lanes' counts, one LDS atomic per wave
    int inc = n_live;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(inc, d);
      if ((tid & 63) >= d) inc += up;
    }
    uint32_t first = 0;
    const int total = __shfl(inc, 63);
    if ((tid & 63) == 63 && total) first = atomicAdd(&cnt, (uint32_t)total);
    first = (uint32_t)__shfl((int)first, 63);
    uint32_t pos = first + (uint32_t)(inc - n_live);
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (w[u] > 0.f) live[pos++] = (uint32_t)(i0 + u);
  }
  __syncthreads();
  const uint32_t n = cnt;
  if (tid == 0) base_s = n ? atomicAdd(reinterpret_cast<uint32_t*>(rows + 3), n) : 0u;   // row 0, slot 12: the list's length
  __syncthreads();
  const uint32_t b = base_s;
  for (uint32_t j = (uint32_t)tid; j < n; j += 1024u) *lr_pbwd_list_slot(rows, b + j) = live[j];
}

template <bool ACCUMULATE, bool SINKROWS>
__global__ void __launch_bounds__(256)
lr_pbwd_list_kernel(LrView v, int N, const float* __restrict__ means, const float* __restrict__ scales,
                    const float* __restrict__ rots, float4* __restrict__ rows, float* __restrict__ o_mean2d,
                    float* __restrict__ o_opac, float* __restrict__ o_col, float* __restrict__ g_means3d,
                    float* __restrict__ g_scales, float* __restrict__ g_rots) {
  const uint32_t n = min(reinterpret_cast<const uint32_t*>(rows + 3)[0], (uint32_t)N);
  for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < n; j += gridDim.x * 256u) {
    const uint32_t i = *lr_pbwd_list_slot(rows, j);
    if (i < (uint32_t)N)
      lr_pbwd_live_row<ACCUMULATE, false, true, SINKROWS>(v, (int)i, means, scales, rots, nullptr, nullptr, rows, o_mean2d,
                                                          o_opac, o_col, g_means3d, g_scales, g_rots);
  }
}

void lr_launch_zero_floats(float* p, size_t n, hipStream_t s);   // project.hip
void lr_launch_zero_words(uint32_t* p, size_t words, hipStream_t s);   // project.hip
// rows != NULL: the 64-byte accumulator rows of lograst_backward (+ its three separate outputs); NULL: g_mean2d / g_conic
void lr_launch_project_bwd(const LrView& v, int N, const float* means, const float* scales, const float* rots,
                           const int* radii, const float* g_mean2d, const float* g_conic, const float* rows,
                           float* o_mean2d, float* o_opac, float* o_col, const float* pw,
                           float* g_means3d, float* g_scales, float* g_rots, bool accumulate, bool sink_rows,
                           hipStream_t s) {
  if (N <= 0) return;
  lr_prof_begin(LRK_PROJECT_BWD, s);
  const dim3 grid((N + LR_PBWD_ROWS - 1) / LR_PBWD_ROWS), block(256);
  const float4* rows4 = reinterpret_cast<const float4*>(rows);
  // dL/dmeans2D is a per-view output, defined for every row.  When the other gradients are running sums (nothing else is
  // written for a dead row) and the input is large, its zeros are streamed by lr_zero_floats_kernel (non-temporal 16-byte
  // stores) in front of the chain rule instead of by its workgroups between their flag reads and their barrier: a band
  // view of 100 M rows 1040 -> 730 us (with the radii read gone), the 30 M view unchanged (415 us).  With fresh gradients
  // (every output zeroed for every dead row: 68 bytes per row) the same split LOSES (30 M: 763 -> 858 us): kept inside.
  LR_KNOB(separate_min_n, "LOGRAST_HELPER_MIN_N", 4000000);
  const int clear_inside = (rows && (accumulate || sink_rows) && N >= separate_min_n) ? 0 : 1;
  if (!clear_inside) lr_launch_zero_floats(o_mean2d, 3 * (size_t)N, s);
  // Large inputs, running sums, the forward's point_weight at hand, no cov3d_precomp: the live rows through a compact list
  // (see lr_pbwd_compact_kernel) -- on BAND views, where a per cent of the rows is live and the one-kernel form is nearly
  // all flag pass (100 M rows, band 3 of 8: 929 -> 490 us).  On full views the chain rule is bound by its scattered lines
  // either way (~7 lines per live row at ~50 G lines/s): measured at 30 M rows, list / one kernel: 7 % live 419 / 469 us,
  // 7.5 % (trained-like) 451 / 481, 14 % (opacity = rand) 690 / 610 -- not worth a second form there.
  // LOGRAST_PBWD_LIST: 0 never, 1 band views (default), 2 always.
  LR_KNOB(list_knob, "LOGRAST_PBWD_LIST", 1);
  const bool band_view = v.ty0 > 0 || v.ty1 < v.gy;
  if ((list_knob == 2 || (list_knob == 1 && band_view)) && !clear_inside && pw && !v.cov3d && N >= 8) {
    float4* rows_w = const_cast<float4*>(rows4);   // (the accumulator rows are the caller's scratch: slots 12-15 are this path's)
    lr_launch_zero_words(reinterpret_cast<uint32_t*>(rows_w + 3), 4, s);
    hipLaunchKernelGGL(lr_pbwd_compact_kernel, dim3((N + LR_PBWD_CHUNK - 1) / LR_PBWD_CHUNK), dim3(1024), 0, s, pw, N, rows_w);
    const dim3 lgrid(2048);
    if (sink_rows)
      hipLaunchKernelGGL((lr_pbwd_list_kernel<true, true>), lgrid, block, 0, s, v, N, means, scales, rots, rows_w, o_mean2d,
                         o_opac, o_col, g_means3d, g_scales, g_rots);
    else
      hipLaunchKernelGGL((lr_pbwd_list_kernel<true, false>), lgrid, block, 0, s, v, N, means, scales, rots, rows_w, o_mean2d,
                         o_opac, o_col, g_means3d, g_scales, g_rots);
    lr_prof_end(LRK_PROJECT_BWD, s);
    return;
  }
  if (sink_rows) {   // (lograst_backward checked: rows != NULL, no cov3d)
    const bool band = v.ty0 > 0 || v.ty1 < v.gy;
    if (pw && band)
      hipLaunchKernelGGL((lr_project_bwd_kernel<true, true, false, true, true, 4>), grid, block, 0, s, v, N, means, scales, rots,
                         radii, g_mean2d, g_conic, rows4, o_mean2d, o_opac, o_col, pw, g_means3d, g_scales, g_rots, clear_inside);
    else if (pw)
      hipLaunchKernelGGL((lr_project_bwd_kernel<true, true, false, true, true>), grid, block, 0, s, v, N, means, scales, rots,
                         radii, g_mean2d, g_conic, rows4, o_mean2d, o_opac, o_col, pw, g_means3d, g_scales, g_rots, clear_inside);
    else
      hipLaunchKernelGGL((lr_project_bwd_kernel<true, false, false, true, true>), grid, block, 0, s, v, N, means, scales, rots,
                         radii, g_mean2d, g_conic, rows4, o_mean2d, o_opac, o_col, pw, g_means3d, g_scales, g_rots, clear_inside);
    lr_prof_end(LRK_PROJECT_BWD, s);
    return;
  }
#define LR_PBWD2(A, T, C, O) hipLaunchKernelGGL((lr_project_bwd_kernel<A, T, C, O>), grid, block, 0, s, v, N, means, scales, \
                                                rots, radii, g_mean2d, g_conic, rows4, o_mean2d, o_opac, o_col, pw,       \
                                                g_means3d, g_scales, g_rots, clear_inside)
#define LR_PBWD(A, T, C) do { if (rows) LR_PBWD2(A, T, C, true); else LR_PBWD2(A, T, C, false); } while (0)
#define LR_PBWD_C(A, T) do { if (v.cov3d) LR_PBWD(A, T, true); else LR_PBWD(A, T, false); } while (0)
  if (accumulate) { if (pw) LR_PBWD_C(true, true); else LR_PBWD_C(true, false); }
  else { if (pw) LR_PBWD_C(false, true); else LR_PBWD_C(false, false); }
#undef LR_PBWD_C
#undef LR_PBWD
#undef LR_PBWD2
  lr_prof_end(LRK_PROJECT_BWD, s);
}

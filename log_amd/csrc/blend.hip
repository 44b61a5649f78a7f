// blend.hip -- A5/A7/A8 forward compositing and A6 backward reverse walk.
//
// CDNA4 mapping (not the 16x16-threads-per-tile CUDA shape): ONE 64-lane wave owns one 16x16 tile and
// every lane owns 4 pixels, one in each 8x8 quadrant.  Per 64 list entries the wave gathers 64 projected
// records into VGPRs (one per lane, ids read coalesced from the sorted list), then walks them with
// v_readlane broadcasts: the current Gaussian lives in SGPRs, so it costs no LDS traffic, no barrier and
// no VGPRs, and each broadcast is amortised over 4 pixels per lane.  A single-wave workgroup needs no
// __syncthreads, terminates exactly when its own 256 pixels are done, and 8160 (1080p) independent
// waves give the dispatcher enough slack to balance uneven tile lists.
//
// Backward: same mapping; the per-(tile,Gaussian) gradient is reduced across the wave with DPP row
// shifts/broadcasts and committed with ONE lane's atomics per (tile, Gaussian) -- not one atomic per
// (pixel, Gaussian) as in the third-party kernel.
#include "common.hpp"

// blockIdx -> tile.  mode 0: identity.  mode 1: XCD-banded -- workgroup b runs on XCD b%8 (observed
// dispatch rule), so XCD k gets the contiguous tile range [k*nper, (k+1)*nper): neighbouring tiles share
// Gaussians and therefore share that XCD's private L2.
LR_DEV uint32_t lr_tile_of_block(uint32_t b, uint32_t tiles, int mode) {
  if (mode == 1) {
    uint32_t nper = (tiles + 7u) >> 3;
    return (b & 7u) * nper + (b >> 3);
  }
  return b;
}

template <bool EXTRAS>
__global__ void __launch_bounds__(64)
lr_blend_fwd_kernel(LrView v, const float4* __restrict__ geom, const uint32_t* __restrict__ state,
                    uint32_t tiles, const uint32_t* __restrict__ plist, uint32_t capacity,
                    float* __restrict__ image, float* __restrict__ final_T, int* __restrict__ n_contrib,
                    int* __restrict__ pid, float* __restrict__ pwp, float* __restrict__ pw, int xcd_mode) {
  if (state[LR_HDR_NUM] > capacity) return;
  const uint32_t tile = lr_tile_of_block(blockIdx.x, tiles, xcd_mode);
  if (tile >= tiles) return;
  const uint32_t* offsets = state + lr_offsets_off(tiles);
  const uint32_t beg = offsets[tile], end = offsets[tile + 1];
  const int lane = threadIdx.x;
  const int tx = tile % (uint32_t)v.gx, ty = tile / (uint32_t)v.gx;
  const int bx = tx * 16 + (lane & 7), by = ty * 16 + (lane >> 3);

  float pxf[4], pyf[4], T[4], C0[4], C1[4], C2[4], wmax[4];
  int wid[4], last[4];
  bool done[4], inside[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    int px = bx + (q & 1) * 8, py = by + (q >> 1) * 8;
    pxf[q] = (float)px; pyf[q] = (float)py;
    inside[q] = (px < v.W) && (py < v.H);
    done[q] = !inside[q];
    T[q] = 1.f; C0[q] = 0.f; C1[q] = 0.f; C2[q] = 0.f; wmax[q] = 0.f; wid[q] = -1; last[q] = 0;
  }

  for (uint32_t base = beg; base < end; base += 64) {
    if (__all(done[0] && done[1] && done[2] && done[3])) break;
    const int cnt = (int)min(64u, end - base);
    uint32_t id = 0;
    float4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
    float cb = 0.f;
    if (lane < cnt) {
      id = plist[base + lane];
      g0 = geom[3 * (size_t)id];
      g1 = geom[3 * (size_t)id + 1];
      cb = reinterpret_cast<const float*>(geom)[12 * (size_t)id + 8];
    }
    const float hA = -0.5f * g0.z, nB = -g0.w, hC = -0.5f * g1.x;
    const int pos0 = (int)(base - beg);
    for (int j = 0; j < cnt; j++) {
      if ((j & 7) == 0 && j && __all(done[0] && done[1] && done[2] && done[3])) break;
      const float mx = lr_readlane_f(g0.x, j), my = lr_readlane_f(g0.y, j);
      const float a = lr_readlane_f(hA, j), b = lr_readlane_f(nB, j), c = lr_readlane_f(hC, j);
      const float op = lr_readlane_f(g1.y, j);
      const float cr = lr_readlane_f(g1.z, j), cg = lr_readlane_f(g1.w, j), cbl = lr_readlane_f(cb, j);
      const int gid = lr_readlane_i((int)id, j);
      float wbest = 0.f;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (!done[q]) {
          const float dx = mx - pxf[q], dy = my - pyf[q];
          const float power = lr_power(a, b, c, dx, dy);
          if (!(power > 0.f)) {
            const float alpha = fminf(0.99f, op * lr_exp(power));
            if (!(alpha < 1.0f / 255.0f)) {
              const float test_T = T[q] * (1.f - alpha);
              if (test_T < 0.0001f) {
                done[q] = true;
              } else {
                const float w = alpha * T[q];
                C0[q] = lr_fma(cr, w, C0[q]); C1[q] = lr_fma(cg, w, C1[q]); C2[q] = lr_fma(cbl, w, C2[q]);
                if (w > wmax[q]) { wmax[q] = w; wid[q] = gid; }
                wbest = fmaxf(wbest, w);
                T[q] = test_T;
                last[q] = pos0 + j + 1;
              }
            }
          }
        }
      }
      if (EXTRAS) {
        if (__any(wbest > 0.f)) {
          const float m = lr_wave_max_to63(wbest);
          if (lane == 63) atomicMax(reinterpret_cast<unsigned int*>(pw) + gid, __float_as_uint(m));
        }
      }
    }
  }

  const size_t plane = (size_t)v.W * v.H;
  const float bg0 = v.bg[0], bg1 = v.bg[1], bg2 = v.bg[2];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    if (inside[q]) {
      const size_t pix = (size_t)(by + (q >> 1) * 8) * v.W + (bx + (q & 1) * 8);
      final_T[pix] = T[q];
      n_contrib[pix] = last[q];
      image[pix] = lr_fma(T[q], bg0, C0[q]);
      image[plane + pix] = lr_fma(T[q], bg1, C1[q]);
      image[2 * plane + pix] = lr_fma(T[q], bg2, C2[q]);
      if (EXTRAS) { pid[pix] = wid[q]; pwp[pix] = wmax[q]; }
    }
  }
}

void lr_launch_blend_fwd(const LrView& v, const void* geom, const uint32_t* state, uint32_t tiles,
                         const uint32_t* plist, uint32_t capacity, float* image, float* final_T, int* n_contrib,
                         int* pid, float* pwp, float* pw, hipStream_t s) {
  static const int xcd_mode = lr_env_int("LOGRAST_XCD_MODE", 1);
  uint32_t grid = xcd_mode == 1 ? ((tiles + 7u) / 8u) * 8u : tiles;
  lr_prof_begin(LRK_BLEND_FWD, s);
  if (v.extras)
    hipLaunchKernelGGL(lr_blend_fwd_kernel<true>, dim3(grid), dim3(64), 0, s, v, reinterpret_cast<const float4*>(geom),
                       state, tiles, plist, capacity, image, final_T, n_contrib, pid, pwp, pw, xcd_mode);
  else
    hipLaunchKernelGGL(lr_blend_fwd_kernel<false>, dim3(grid), dim3(64), 0, s, v, reinterpret_cast<const float4*>(geom),
                       state, tiles, plist, capacity, image, final_T, n_contrib, pid, pwp, pw, xcd_mode);
  lr_prof_end(LRK_BLEND_FWD, s);
}

// ---- backward ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
lr_blend_bwd_kernel(LrView v, const float4* __restrict__ geom, const uint32_t* __restrict__ state,
                    uint32_t tiles, const uint32_t* __restrict__ plist, uint32_t capacity,
                    const float* __restrict__ final_T, const int* __restrict__ n_contrib,
                    const float* __restrict__ dL_dimage, float* __restrict__ g_mean2d,
                    float* __restrict__ g_conic, float* __restrict__ g_opac, float* __restrict__ g_col,
                    int xcd_mode) {
  if (state[LR_HDR_NUM] > capacity) return;
  const uint32_t tile = lr_tile_of_block(blockIdx.x, tiles, xcd_mode);
  if (tile >= tiles) return;
  const uint32_t* offsets = state + lr_offsets_off(tiles);
  const uint32_t beg = offsets[tile];
  const int lane = threadIdx.x;
  const int tx = tile % (uint32_t)v.gx, ty = tile / (uint32_t)v.gx;
  const int bx = tx * 16 + (lane & 7), by = ty * 16 + (lane >> 3);
  const size_t plane = (size_t)v.W * v.H;
  const float bg0 = v.bg[0], bg1 = v.bg[1], bg2 = v.bg[2];
  const float sx = 0.5f * (float)v.W, sy = 0.5f * (float)v.H;

  float pxf[4], pyf[4], T[4], Tf[4], dp0[4], dp1[4], dp2[4], bgdot[4];
  float acc0[4], acc1[4], acc2[4], lal[4], lc0[4], lc1[4], lc2[4];
  int lastc[4];
  int maxc = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    int px = bx + (q & 1) * 8, py = by + (q >> 1) * 8;
    pxf[q] = (float)px; pyf[q] = (float)py;
    bool in = (px < v.W) && (py < v.H);
    size_t pix = in ? (size_t)py * v.W + px : 0;
    Tf[q] = in ? final_T[pix] : 0.f;
    lastc[q] = in ? n_contrib[pix] : 0;
    dp0[q] = in ? dL_dimage[pix] : 0.f;
    dp1[q] = in ? dL_dimage[plane + pix] : 0.f;
    dp2[q] = in ? dL_dimage[2 * plane + pix] : 0.f;
    bgdot[q] = lr_fma(bg0, dp0[q], lr_fma(bg1, dp1[q], bg2 * dp2[q]));
    T[q] = Tf[q];
    acc0[q] = acc1[q] = acc2[q] = 0.f; lal[q] = 0.f; lc0[q] = lc1[q] = lc2[q] = 0.f;
    maxc = max(maxc, lastc[q]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) maxc = max(maxc, __shfl_xor(maxc, off));
  maxc = lr_readlane_i(maxc, 0);  // wave-uniform

  for (int hi = maxc; hi > 0; hi -= 64) {
    const int cnt = min(64, hi);
    uint32_t id = 0;
    float4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
    float cb = 0.f;
    if (lane < cnt) {
      id = plist[beg + (uint32_t)(hi - 1 - lane)];
      g0 = geom[3 * (size_t)id];
      g1 = geom[3 * (size_t)id + 1];
      cb = reinterpret_cast<const float*>(geom)[12 * (size_t)id + 8];
    }
    const float hA = -0.5f * g0.z, nB = -g0.w, hC = -0.5f * g1.x;
    for (int j = 0; j < cnt; j++) {
      const int k = hi - 1 - j;  // 0-based position in the tile list
      const float mx = lr_readlane_f(g0.x, j), my = lr_readlane_f(g0.y, j);
      const float a = lr_readlane_f(hA, j), b = lr_readlane_f(nB, j), c = lr_readlane_f(hC, j);
      const float op = lr_readlane_f(g1.y, j);
      const float cr = lr_readlane_f(g1.z, j), cg = lr_readlane_f(g1.w, j), cbl = lr_readlane_f(cb, j);
      const int gid = lr_readlane_i((int)id, j);
      float s_c0 = 0.f, s_c1 = 0.f, s_c2 = 0.f, s_mx = 0.f, s_my = 0.f, s_A = 0.f, s_B = 0.f, s_C = 0.f, s_op = 0.f;
      bool hit = false;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (k < lastc[q]) {
          const float dx = mx - pxf[q], dy = my - pyf[q];
          const float power = lr_power(a, b, c, dx, dy);
          if (!(power > 0.f)) {
            const float G = lr_exp(power);
            const float alpha = fminf(0.99f, op * G);
            if (!(alpha < 1.0f / 255.0f)) {
              const float om = 1.f - alpha;
              float rc = __builtin_amdgcn_rcpf(om);
              rc = lr_fma(lr_fma(-om, rc, 1.f), rc, rc);  // one Newton step: v_rcp_f32's 1-ulp bias compounds over long lists
              T[q] = T[q] * rc;
              const float w = alpha * T[q];
              acc0[q] = lr_fma(lal[q], lc0[q], (1.f - lal[q]) * acc0[q]);
              acc1[q] = lr_fma(lal[q], lc1[q], (1.f - lal[q]) * acc1[q]);
              acc2[q] = lr_fma(lal[q], lc2[q], (1.f - lal[q]) * acc2[q]);
              lc0[q] = cr; lc1[q] = cg; lc2[q] = cbl;
              float dL_dalpha = lr_fma(cr - acc0[q], dp0[q], lr_fma(cg - acc1[q], dp1[q], (cbl - acc2[q]) * dp2[q]));
              dL_dalpha = lr_fma(dL_dalpha, T[q], -(Tf[q] * rc) * bgdot[q]);
              lal[q] = alpha;
              const float dL_dG = op * dL_dalpha;
              const float gdx = G * dx, gdy = G * dy;
              const float dG_ddx = lr_fma(2.f * a, gdx, b * gdy);   // -gdx*A - gdy*B
              const float dG_ddy = lr_fma(2.f * c, gdy, b * gdx);   // -gdy*C - gdx*B
              s_c0 = lr_fma(w, dp0[q], s_c0); s_c1 = lr_fma(w, dp1[q], s_c1); s_c2 = lr_fma(w, dp2[q], s_c2);
              s_mx = lr_fma(dL_dG, dG_ddx, s_mx); s_my = lr_fma(dL_dG, dG_ddy, s_my);
              s_A = lr_fma(-0.5f * gdx * dx, dL_dG, s_A);
              s_B = lr_fma(-gdx * dy, dL_dG, s_B);
              s_C = lr_fma(-0.5f * gdy * dy, dL_dG, s_C);
              s_op = lr_fma(G, dL_dalpha, s_op);
              hit = true;
            }
          }
        }
      }
      if (__any(hit)) {
        s_c0 = lr_wave_sum_to63(s_c0); s_c1 = lr_wave_sum_to63(s_c1); s_c2 = lr_wave_sum_to63(s_c2);
        s_mx = lr_wave_sum_to63(s_mx); s_my = lr_wave_sum_to63(s_my);
        s_A = lr_wave_sum_to63(s_A); s_B = lr_wave_sum_to63(s_B); s_C = lr_wave_sum_to63(s_C);
        s_op = lr_wave_sum_to63(s_op);
        if (lane == 63) {
          atomicAdd(g_col + 3 * (size_t)gid + 0, s_c0);
          atomicAdd(g_col + 3 * (size_t)gid + 1, s_c1);
          atomicAdd(g_col + 3 * (size_t)gid + 2, s_c2);
          atomicAdd(g_mean2d + 3 * (size_t)gid + 0, s_mx * sx);
          atomicAdd(g_mean2d + 3 * (size_t)gid + 1, s_my * sy);
          atomicAdd(g_conic + 4 * (size_t)gid + 0, s_A);
          atomicAdd(g_conic + 4 * (size_t)gid + 1, s_B);
          atomicAdd(g_conic + 4 * (size_t)gid + 2, s_C);
          atomicAdd(g_opac + gid, s_op);
        }
      }
    }
  }
}

void lr_launch_blend_bwd(const LrView& v, const void* geom, const uint32_t* state, uint32_t tiles,
                         const uint32_t* plist, uint32_t capacity, const float* final_T, const int* n_contrib,
                         const float* dL_dimage, float* g_mean2d, float* g_conic, float* g_opac, float* g_col,
                         hipStream_t s) {
  static const int xcd_mode = lr_env_int("LOGRAST_XCD_MODE", 1);
  uint32_t grid = xcd_mode == 1 ? ((tiles + 7u) / 8u) * 8u : tiles;
  lr_prof_begin(LRK_BLEND_BWD, s);
  hipLaunchKernelGGL(lr_blend_bwd_kernel, dim3(grid), dim3(64), 0, s, v, reinterpret_cast<const float4*>(geom), state,
                     tiles, plist, capacity, final_T, n_contrib, dL_dimage, g_mean2d, g_conic, g_opac, g_col,
                     xcd_mode);
  lr_prof_end(LRK_BLEND_BWD, s);
}

// project.hip -- per-Gaussian streaming kernels: A0 compute_radius, A1 projection (+ A2 tile counting),
// tile scan, and A3 per-tile bucket fill.  HBM-bound: every Gaussian attribute is read exactly once,
// coalesced; the 64-byte projected record is written once.
#include "common.hpp"

// ---- A0 -------------------------------------------------------------------------------------------
// Stands for compute_radius_cuda (/root/reference/LoG/cuda/compute_radius_kernel.cu:107-156):
// float radius (no ceil), |ndc|>1.3 cull only, fork low-pass max(.,0.3), det==0 -> 0.
__global__ void __launch_bounds__(256)
lr_radius_kernel(int P, const float* __restrict__ means, const float* __restrict__ scales,
                 const float* __restrict__ rots, const float* __restrict__ proj,
                 const float* __restrict__ view, float fx, float fy, float tanfovx, float tanfovy,
                 float* __restrict__ radii) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  // all 44 input bytes are requested up front (three independent vector loads in flight per lane); the cull
  // below only decides whether the arithmetic runs
  float p[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
  float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
  const float4 q4 = reinterpret_cast<const float4*>(rots)[i];
  const float q[4] = {q4.x, q4.y, q4.z, q4.w};
  const float out = lr_radius_one(p, s, q, proj, view, fx, fy, tanfovx, tanfovy);
  radii[i] = out;
}

void lr_launch_radius(int P, const float* means, const float* scales, const float* rots, const float* proj,
                      const float* view, float fx, float fy, float tanfovx, float tanfovy, float* radii,
                      hipStream_t s) {
  if (P <= 0) return;
  lr_prof_begin(LRK_RADIUS, s);
  hipLaunchKernelGGL(lr_radius_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means, scales, rots, proj,
                     view, fx, fy, tanfovx, tanfovy, radii);
  lr_prof_end(LRK_RADIUS, s);
}

// ---- A1 + A2 ----------------------------------------------------------------------------------------
// One thread per Gaussian.  Writes radii[i] (0 = culled) and the 64-byte record
//   q0 = (mx, my, conicA, conicB)  q1 = (conicC, opacity, r, g)  q2 = (b, depth, rect_min, rect_max)
//   q3 = slots of its (<= 4) tile instances inside their tiles, row-major over the rect
// (q2 is written for every Gaussian, zero = culled/empty rect).  Counting and slot assignment are one
// returning atomic per instance, so the bucket fill needs no atomics for these Gaussians; larger rects are
// only counted here (non-returning atomic) and placed by the fill kernel.
// Support cull (tile_cull): a tile of the rect becomes an instance only if the Gaussian can reach alpha >= 1/255
// somewhere inside it (lr_support_tile, conservative).  The rect is the bounding square of the 3-sigma circle of
// the LARGER eigenvalue, so for elongated splats a good part of its tiles can never contribute; dropping them
// changes no output (they fail the alpha floor at every pixel) but shortens every list that is counted, filled,
// sorted and walked.  radii[] keeps the reference meaning (rect non-empty).
// Counter policies: where the per-tile counters that rank / count the tile instances live.
struct LrGlobalCounters {  // one memory-side atomic per instance (any tile grid)
  uint32_t* ranked;
  uint32_t* big;
  LR_DEV uint32_t rank(int tile) const { return atomicAdd(&ranked[tile * LR_CTR_STRIDE], 1u); }
  LR_DEV void count_big(int tile) const { atomicAdd(&big[tile * LR_CTR_STRIDE], 1u); }
};
struct LrLdsCounters {  // per-workgroup counters in LDS (batched kernel): 170x the rate of memory-side atomics
  uint32_t* ctr;        // one word per tile: ranked count in the low half, big count in the high half (both < 2^16:
                        // a batch holds at most 32768 Gaussians and a Gaussian enters a tile at most once)
  LR_DEV uint32_t rank(int tile) const { return atomicAdd(&ctr[tile], 1u) & 0xffffu; }
  LR_DEV void count_big(int tile) const { atomicAdd(&ctr[tile], 0x10000u); }
};

// The 56 input bytes of one Gaussian, requested together (five independent loads in flight per lane).
struct LrInputs {
  float p[3], s[3], c[3];
  float4 q;
  float op;
};
// (cov3d: the rasterizer's cov3D_precomp instead of scales + rotations -- its six floats travel in s[] and q.xyz, so the
// prefetch holds no more registers)
// RECT_ONLY: opacity and colour are left out (16 of the 56 bytes) -- what lr_project_rect needs; lr_project_band_kernel
// fetches the other two for the Gaussians that turn out to have a rect.
template <bool RECT_ONLY = false>
LR_DEV LrInputs lr_load_inputs(int i, const float* __restrict__ means, const float* __restrict__ scales,
                               const float* __restrict__ rots, const float* __restrict__ opac,
                               const float* __restrict__ colors, const float* __restrict__ cov3d) {
  LrInputs in;
  in.p[0] = means[3 * i]; in.p[1] = means[3 * i + 1]; in.p[2] = means[3 * i + 2];
  if (cov3d) {
    const float* __restrict__ c6 = cov3d + 6 * (size_t)i;
    in.s[0] = c6[0]; in.s[1] = c6[1]; in.s[2] = c6[2];
    in.q = float4{c6[3], c6[4], c6[5], 0.f};
  } else {
    in.s[0] = scales[3 * i]; in.s[1] = scales[3 * i + 1]; in.s[2] = scales[3 * i + 2];
    in.q = reinterpret_cast<const float4*>(rots)[i];
  }
  if (RECT_ONLY) {
    in.op = 0.f; in.c[0] = 0.f; in.c[1] = 0.f; in.c[2] = 0.f;
  } else {
    in.op = opac[i];
    in.c[0] = colors[3 * i]; in.c[1] = colors[3 * i + 1]; in.c[2] = colors[3 * i + 2];
  }
  return in;
}

// The rect of one Gaussian: everything lr_project_one computes from means3D / scales / rotations alone.
struct LrRect {
  float mx, my, cA, cB, cC, tz;
  int x0, y0, x1, y1, rad;
};
LR_DEV bool lr_project_rect(const LrView& v, const LrInputs& in, LrRect& r) {
  const lr_cfloat* V = lr_uniform(v.view);
  const lr_cfloat* Pm = lr_uniform(v.proj);
  const float p[3] = {in.p[0], in.p[1], in.p[2]};
  float tz = lr_dot3p(V[2], V[6], V[10], p[0], p[1], p[2], V[14]);
  if (!(tz > 0.2f)) return false;
  float hx = lr_dot3p(Pm[0], Pm[4], Pm[8], p[0], p[1], p[2], Pm[12]);
  float hy = lr_dot3p(Pm[1], Pm[5], Pm[9], p[0], p[1], p[2], Pm[13]);
  float hw = lr_dot3p(Pm[3], Pm[7], Pm[11], p[0], p[1], p[2], Pm[15]);
  float pw = 1.0f / (hw + 0.0000001f);
  float nx = hx * pw, ny = hy * pw;
  if (v.ndc_cull && (nx < -1.3f || nx > 1.3f || ny < -1.3f || ny > 1.3f)) return false;
  float Sg[6];
  if (v.cov3d) {   // cov3D_precomp (wave-uniform): see lr_load_inputs
    Sg[0] = in.s[0]; Sg[1] = in.s[1]; Sg[2] = in.s[2]; Sg[3] = in.q.x; Sg[4] = in.q.y; Sg[5] = in.q.z;
  } else {
    float s[3] = {in.s[0] * v.scale_modifier, in.s[1] * v.scale_modifier, in.s[2] * v.scale_modifier};
    float q[4] = {in.q.x, in.q.y, in.q.z, in.q.w};
    float R[9];
    lr_cov3d(s, q, R, Sg);
  }
  LrEwa e;
  lr_ewa(p, Sg, V, v.fx, v.fy, v.tanfovx, v.tanfovy, v.filter_mode, e);
  float det = e.a * e.c - e.b * e.b;
  if (det == 0.0f) return false;
  float det_inv = 1.f / det;
  r.cA = e.c * det_inv; r.cB = -e.b * det_inv; r.cC = e.a * det_inv;
  float rf = ceilf(lr_radius_from_cov(e.a, e.c, det));
  float mx = ((nx + 1.0f) * (float)v.W - 1.0f) * 0.5f;
  float my = ((ny + 1.0f) * (float)v.H - 1.0f) * 0.5f;
  if (!((rf <= 1048576.f) && (fabsf(mx) < 1.0e8f) && (fabsf(my) < 1.0e8f))) return false;
  int x0 = (int)((mx - rf) / 16.f), y0 = (int)((my - rf) / 16.f);
  int x1 = (int)(((mx + rf) + 15.f) / 16.f), y1 = (int)(((my + rf) + 15.f) / 16.f);
  x0 = min(v.gx, max(0, x0)); x1 = min(v.gx, max(0, x1));
  y0 = min(v.ty1, max(v.ty0, y0)); y1 = min(v.ty1, max(v.ty0, y1));   // [ty0, ty1) = [0, gy) unless the image is split
  if ((x1 - x0) * (y1 - y0) <= 0) return false;
  r.mx = mx; r.my = my; r.tz = tz; r.x0 = x0; r.y0 = y0; r.x1 = x1; r.y1 = y1; r.rad = (int)rf;
  return true;
}

// Projection of one Gaussian (A1) + counting / ranking of its tile instances (A2) against `ctr`.  Outputs the four
// record quads, the integer radius (0 = culled) and the rect-rule instance count.
// DEFER_HUGE: rects of more than `defer_tiles` tiles are not counted here (`huge` is raised instead and
// lr_count_huge_kernel counts them, one wave per rect) -- a lane walking an 81-tile rect with a 50-instruction support
// test per tile holds up its whole wave, and in level-of-detail order the big splats sit together in a few batches.
template <bool DEFER_HUGE, typename Counters>
LR_DEV void lr_project_one(const LrView& v, const LrInputs& in, int tile_cull, const Counters& ctr, float4& g0,
                           float4& g1, float4& g2, float4& g3, int& rad, uint32_t& rect_instances, bool& huge,
                           int defer_tiles = LR_COOP_TILES) {
  huge = false;
  rad = 0;
  g0 = float4{0.f, 0.f, 0.f, 0.f};
  g1 = g0; g3 = g0;
  g2 = g0;  // culled: empty rect (the fill kernel reads only q2)
  LrRect rc;
  if (!lr_project_rect(v, in, rc)) return;
  const float mx = rc.mx, my = rc.my, cA = rc.cA, cB = rc.cB, cC = rc.cC, tz = rc.tz;
  const int x0 = rc.x0, y0 = rc.y0, x1 = rc.x1, y1 = rc.y1;
  rad = rc.rad;
  g0 = float4{mx, my, cA, cB};
  g1 = float4{cC, in.op, in.c[0], in.c[1]};
  g2 = float4{in.c[2], tz, __uint_as_float((uint32_t)x0 | ((uint32_t)y0 << 16)),
              __uint_as_float((uint32_t)x1 | ((uint32_t)y1 << 16))};
  const int w = x1 - x0, nt = w * (y1 - y0);
  rect_instances += (uint32_t)nt;
  const LrSupport sup = lr_support_prepare(mx, my, cA, cB, cC, g1.y);
  if (nt <= LR_RANKED_TILES) {
    uint32_t slot[LR_RANKED_TILES] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < LR_RANKED_TILES; k++) {
      if (k < nt) {
        // nt <= 4: the rect is one row (w >= nt), one column (w == 1) or 2x2 -- no integer division
        const int ty = (w == 1) ? k : ((w == 2 && nt == 4) ? (k >> 1) : 0), tx = k - ty * w;
        // a single-tile rect holds the centre's neighbourhood: testing it would almost never drop it
        const bool keep = !tile_cull || nt == 1 || lr_support_tile(sup, x0 + tx, y0 + ty);
        slot[k] = keep ? ctr.rank((y0 + ty) * v.gx + (x0 + tx)) : 0xffffffffu;
      }
    }
    g3 = float4{__uint_as_float(slot[0]), __uint_as_float(slot[1]), __uint_as_float(slot[2]),
                __uint_as_float(slot[3])};
  } else if (DEFER_HUGE && nt > defer_tiles) {
    huge = true;
  } else {
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++)
        if (!tile_cull || lr_support_tile(sup, x, y)) ctr.count_big(y * v.gx + x);
  }
}

// rect-rule instance count (reporting only): one atomic per WORKGROUP -- one per wave (15 K same-address atomics
// at 1 M Gaussians) cost a full-grid launch 40 us
template <int WAVES>
LR_DEV void lr_commit_rect_count(uint32_t rect_instances, uint32_t* hdr) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) rect_instances += (uint32_t)__shfl_xor((int)rect_instances, d);
  __shared__ uint32_t rect_part[WAVES];
  if ((threadIdx.x & 63) == 0) rect_part[threadIdx.x >> 6] = rect_instances;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < WAVES; k++) r += rect_part[k];
    if (r) atomicAdd(&hdr[LR_HDR_RECT], r);
  }
}

// Unbatched kernel (any tile grid): counters in memory, one returning atomic per ranked instance.
__global__ void __launch_bounds__(256)
lr_project_kernel(LrView v, int N, const float* __restrict__ means, const float* __restrict__ scales,
                  const float* __restrict__ rots, const float* __restrict__ opac,
                  const float* __restrict__ colors, int* __restrict__ radii, float4* __restrict__ geom,
                  uint32_t* __restrict__ ranked, uint32_t* __restrict__ big, uint32_t* __restrict__ hdr,
                  int tile_cull) {
  // Records leave through LDS: a lane's four quads are 64 B apart from its neighbour's, so storing them
  // directly makes every store instruction touch 64 different lines with 16 B each.  Staged, each instruction
  // writes 1 KB of contiguous memory.
  __shared__ float4 stage[256 * LR_REC_QUADS];
  float4* wstage = stage + (threadIdx.x & ~63) * LR_REC_QUADS;
  const int lane = threadIdx.x & 63;
  uint32_t rect_instances = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { hdr[LR_HDR_CULL] = tile_cull ? 1u : 0u; hdr[LR_HDR_BATCH] = 0u; }
  const LrGlobalCounters ctr{ranked, big};
  // Grid-stride: the kernel is bound by memory-side atomic throughput (measured: ~60 us without its atomics,
  // ~180 us with them = 24 G atomics/s, the rate a bare atomic microbenchmark reaches on random tile counters),
  // which a few hundred waves in flight already saturate.
  for (int i0 = blockIdx.x * 256 + (threadIdx.x & ~63); i0 < N; i0 += gridDim.x * 256) {  // i0: the wave's first Gaussian
    const int i = i0 + lane;
    float4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, g2 = g0, g3 = g0;
    if (i < N) {
      int rad;
      const LrInputs in = lr_load_inputs(i, means, scales, rots, opac, colors, v.cov3d);
      bool huge;
      lr_project_one<false>(v, in, tile_cull, ctr, g0, g1, g2, g3, rad, rect_instances, huge);
      radii[i] = rad;
    }
    wstage[lane * LR_REC_QUADS + 0] = g0;
    wstage[lane * LR_REC_QUADS + 1] = g1;
    wstage[lane * LR_REC_QUADS + 2] = g2;
    wstage[lane * LR_REC_QUADS + 3] = g3;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nq = min(64, N - i0) * LR_REC_QUADS;  // quads this wave owns
#pragma unroll
    for (int k = 0; k < LR_REC_QUADS; k++) {
      const int qd = k * 64 + lane;
      if (qd < nq) geom[LR_REC_QUADS * (size_t)i0 + qd] = wstage[qd];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  lr_commit_rect_count<4>(rect_instances, hdr);
}

// Batched kernel (tile grids whose counters fit in LDS): batch b owns Gaussians [b*B, (b+1)*B).  Its instances
// are counted and ranked in LDS counters (integer LDS atomics run at ~4 T/s chip-wide, memory-side atomics at
// 0.025 T/s); afterwards ONE memory-side atomic per non-empty tile reserves a range of slots in that tile, and the
// batch's start inside it goes to basetab[b][tile].  An instance's slot is basetab[batch][tile] + its rank inside the
// batch (kept in the fill record) -- the fill kernel adds the two.  With B >> tiles / (instances per Gaussian) the
// memory-side atomics shrink by the average number of instances a batch puts into a tile.
// A workgroup owns S CONSECUTIVE batches (one plane of LDS counters each, S <= 4: 4 x 8160 words at 1080p) and reserves
// for all of them with one atomic per tile, so their S runs are adjacent in the tile's list: at 30 M Gaussians a
// (batch, tile) run is ~19 keys = 152 B, which the fill kernel's scattered 8-byte stores left as partial lines
// (0.89 GB of write traffic for 0.35 GB of keys); four adjacent runs are 608 B, written by neighbouring workgroups of
// one XCD within microseconds of each other.  It also divides the reservation atomics by S.
// 82 VGPRs: one 1024-thread workgroup per CU.  Forcing two (amdgpu_waves_per_eu(8): 64 VGPRs, 19 spilled) is slower
// at every size (1 M: 62 -> 81 us, 30 M: 1.06 -> 1.68 ms).
// 4x4 transpose of float4 "elements" inside every group of four consecutive lanes: on entry lane m of a group holds
// (a, b, c, d) = its own four values, on exit value k of lane m is what lane k held in slot m.  Two butterfly stages of
// DPP quad permutes (lane ^ 1, lane ^ 2), per dword.
template <int CTRL>
LR_DEV float lr_quad_perm_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
LR_DEV void lr_quad_exchange(float4& lo, float4& hi, bool upper) {
  // lanes with `upper` send lo and keep hi; the others send hi and keep lo; each receives its partner's value in the slot it sent
  const float4 send = upper ? lo : hi;
  const float4 got = {lr_quad_perm_f<CTRL>(send.x), lr_quad_perm_f<CTRL>(send.y), lr_quad_perm_f<CTRL>(send.z),
                      lr_quad_perm_f<CTRL>(send.w)};
  if (upper) lo = got; else hi = got;
}
LR_DEV void lr_quad_transpose(float4& a, float4& b, float4& c, float4& d, int m) {
  lr_quad_exchange<0xB1>(a, b, (m & 1) != 0);   // quad_perm [1,0,3,2]
  lr_quad_exchange<0xB1>(c, d, (m & 1) != 0);
  lr_quad_exchange<0x4E>(a, c, (m & 2) != 0);   // quad_perm [2,3,0,1]
  lr_quad_exchange<0x4E>(b, d, (m & 2) != 0);
}

// Fill record (16 B, its own coalesced array behind the records): everything lr_fill_kernel needs, so that it
// does not fetch half of every 64-byte record again.  x = depth bits; y = x0 | y0<<13 | (w-1)<<26 | (h-1)<<28 |
// big<<30 (all ones = nothing to fill); ranked: z,w = four 16-bit ranks inside the batch (0xffff = tile dropped
// by the support cull); big: z = x1 | y1<<16, and with bit 31 of y: w = the rect's rank row (5..16 tiles, ranked by the
// batched projection: common.hpp LR_MID_ROW).
LR_DEV uint4 lr_fill_record(const float4& g2, const float4& g3, int rad) {
  uint4 fr = {__float_as_uint(g2.y), 0xffffffffu, 0u, 0u};
  if (rad > 0) {
    const uint32_t r0 = __float_as_uint(g2.z), r1 = __float_as_uint(g2.w);
    const uint32_t x0 = r0 & 0xffffu, y0 = r0 >> 16, x1 = r1 & 0xffffu, y1 = r1 >> 16, w = x1 - x0, h = y1 - y0;
    if (w * h <= LR_RANKED_TILES) {
      const uint32_t s0 = __float_as_uint(g3.x), s1 = __float_as_uint(g3.y), s2 = __float_as_uint(g3.z),
                     s3 = __float_as_uint(g3.w);
      fr.y = x0 | (y0 << 13) | ((w - 1u) << 26) | ((h - 1u) << 28);
      fr.z = (s0 & 0xffffu) | (s1 << 16);                  // 0xffffffff -> 0xffff; ranks are < 32768
      fr.w = (s2 & 0xffffu) | (s3 << 16);
    } else {
      fr.y = x0 | (y0 << 13) | (1u << 30);
      fr.z = x1 | (y1 << 16);
    }
  }
  return fr;
}

#define LR_MAX_PLANES 4
// Reservations of a workgroup's batches (the tail of both projection kernels): eight tiles per thread per round, all
// eight returning atomics in flight before the first result is stored (one memory round trip per round, not eight);
// one atomic covers the workgroup's S batches.  Tiles [t_lo, t_hi); the LDS plane of batch pl starts at pl * stride and
// holds tile t at t - t_lo.
LR_DEV void lr_reserve_batches(const uint32_t* lds_ctr, int stride, int t_lo, int t_hi, int nplanes, int tiles,
                               uint32_t* __restrict__ ranked, uint32_t* __restrict__ big, uint32_t* mybase) {
  for (int t0 = t_lo + (int)threadIdx.x; t0 < t_hi; t0 += 8 * LR_BATCH_THREADS) {
    uint32_t base[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int t = t0 + u * LR_BATCH_THREADS;
      uint32_t c = 0u, cb = 0u;
      if (t < t_hi)
        for (int pl = 0; pl < nplanes; pl++) { const uint32_t packed = lds_ctr[pl * stride + (t - t_lo)]; c += packed & 0xffffu; cb += packed >> 16; }
      base[u] = c ? atomicAdd(&ranked[t], c) : 0u;          // dense counters: see lr_scan_kernel
      if (cb) atomicAdd(&big[t], cb);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int t = t0 + u * LR_BATCH_THREADS;
      if (t < t_hi) {
        uint32_t run = base[u];
        for (int pl = 0; pl < nplanes; pl++) { mybase[(size_t)pl * tiles + t] = run; run += lds_ctr[pl * stride + (t - t_lo)] & 0xffffu; }
      }
    }
  }
}

// The hot loop (round 4).  Everything a lane does for its Gaussian is straight-line code behind ONE validity predicate
// (culls clear the predicate instead of branching: in a wave of 64 consecutive Gaussians some lane always survives, so
// the branches only cost exec-mask bookkeeping, phi copies and -- through the masks held live -- scalar-register spills):
//   * the wave's 64 Gaussians are addressed as wave-uniform base (scalar registers) + a lane offset that never changes,
//     so no per-iteration 64-bit address arithmetic on the vector ALU;
//   * the values that need the raw inputs (view-space centre, clip-space centre, world-space covariance) are computed
//     first; the NEXT iteration's means / scales / rotations are then requested into the registers that just died (the
//     software pipeline needs no copies at the loop's back edge); opacity and colour, used last, are requested at the top;
//   * the (up to four) tiles of a small rect are tested against the alpha support two at a time (lr_support_tile2), then
//     ranked with up to four LDS atomics in flight;
//   * records leave as full 64-byte lines through a 4x4 transpose between the wave's four 16-lane ROWS:
//     v_permlane16_swap / v_permlane32_swap exchange two registers' halves in one instruction (16 instructions for the
//     16 dwords; the DPP quad-permute form took ~100).  Afterwards lane (row r, column l) holds quad r of the records of
//     Gaussians l, 16 + l, 32 + l, 48 + l of the wave: every store instruction writes 16 complete lines.
// Stores of the projection's outputs (records: read next by the compositing kernels, a sort later; fill records: by the
// fill kernel, after the scan).  LR_PROJECT_NT_STORES: as non-temporal (streaming) stores.
typedef float lr_f4v __attribute__((ext_vector_type(4)));
typedef uint32_t lr_u4w __attribute__((ext_vector_type(4)));
LR_DEV void lr_out_store(float4* p, const float4& v) {
#ifdef LR_PROJECT_NT_STORES
  __builtin_nontemporal_store(lr_f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<lr_f4v*>(p));
#else
  *p = v;
#endif
}
LR_DEV void lr_out_store(uint4* p, const uint4& v) {
#ifdef LR_PROJECT_NT_STORES
  __builtin_nontemporal_store(lr_u4w{v.x, v.y, v.z, v.w}, reinterpret_cast<lr_u4w*>(p));
#else
  *p = v;
#endif
}
LR_DEV void lr_out_store(int* p, int v) {
#ifdef LR_PROJECT_NT_STORES
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
// s_waitcnt immediate (gfx9 encoding): vmcnt in bits [3:0] + [15:14], expcnt [6:4] and lgkmcnt [11:8] left at "no wait"
#define LR_WAIT_VMCNT(n) ((((n) & 15) | (((n) >> 4) << 14)) | (7 << 4) | (15 << 8))
LR_DEV void lr_swap16(float& a, float& b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
LR_DEV void lr_swap32(float& a, float& b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
LR_DEV void lr_swap16(float4& a, float4& b) { lr_swap16(a.x, b.x); lr_swap16(a.y, b.y); lr_swap16(a.z, b.z); lr_swap16(a.w, b.w); }
LR_DEV void lr_swap32(float4& a, float4& b) { lr_swap32(a.x, b.x); lr_swap32(a.y, b.y); lr_swap32(a.z, b.z); lr_swap32(a.w, b.w); }
// on entry lane (row r, column l) holds quads 0..3 of ITS record in (a, b, c, d); on exit quad r of the records of the
// lanes (0, l), (1, l), (2, l), (3, l)
LR_DEV void lr_row_transpose(float4& a, float4& b, float4& c, float4& d) {
  lr_swap16(a, b); lr_swap16(c, d);     // a = [a0 b0 a2 b2], b = [a1 b1 a3 b3] (subscript: row of origin)
  lr_swap32(a, c); lr_swap32(b, d);     // a = [a0 b0 c0 d0], b = [a1 b1 c1 d1], c = [a2 b2 c2 d2], d = [a3 b3 c3 d3]
}

// The four quads of the wave's 64 records, from one lane = one Gaussian to full 64-byte lines per store instruction.
LR_DEV void lr_store_records(float4* __restrict__ rec, int lane, float4& g0, float4& g1, float4& g2, float4& g3) {
#ifdef LR_PROJECT_QUAD_STORES
  const int m = lane & 3, b = lane - m;
  lr_quad_transpose(g0, g1, g2, g3, m);                    // g<k> = quad m of the record of lane (lane - m + k)
  lr_out_store(&rec[LR_REC_QUADS * (b + 0) + m], g0);
  lr_out_store(&rec[LR_REC_QUADS * (b + 1) + m], g1);
  lr_out_store(&rec[LR_REC_QUADS * (b + 2) + m], g2);
  lr_out_store(&rec[LR_REC_QUADS * (b + 3) + m], g3);
#else
  lr_row_transpose(g0, g1, g2, g3);                        // g<k> = quad (lane >> 4) of the record of Gaussian 16 k + (lane & 15)
  const int l16 = lane & 15, r = lane >> 4;
  lr_out_store(&rec[LR_REC_QUADS * l16 + r], g0);
  lr_out_store(&rec[LR_REC_QUADS * (l16 + 16) + r], g1);
  lr_out_store(&rec[LR_REC_QUADS * (l16 + 32) + r], g2);
  lr_out_store(&rec[LR_REC_QUADS * (l16 + 48) + r], g3);
#endif
}
// means3D / scales / rotations (or the six covariance floats) of one Gaussian: what the first part of the loop consumes
struct LrGeo { float p[3], s[3]; float4 q; };
// A wave-uniform index, pinned to a scalar register and opaque to the loop optimiser -- which otherwise folds the lane's
// (invariant) offset into a VECTOR base and adds the (changing) scalar part per iteration with v_mad_i64: with it the
// accesses are `global_load ... v_lane_offset, s[base:base+1]`.
LR_DEV uint32_t lr_sgpr(uint32_t i) { asm volatile("" : "+s"(i)); return i; }
// ... and a lane offset that is not available before the listed values are: the scheduler otherwise hoists the request
// for the NEXT Gaussian's inputs above the arithmetic that consumes the current ones, and then has to keep two sets of
// inputs alive and copy one into the other at the loop's back edge.
LR_DEV uint32_t lr_after(uint32_t off, const float a[3], float b0, float b1, float b2, const float c[6]) {
  asm volatile("" : "+v"(off) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(b0), "v"(b1), "v"(b2), "v"(c[0]), "v"(c[1]),
               "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]));
  return off;
}
// wm / ws / wr / wc6: the arrays at the wave's first Gaussian (wave-uniform: scalar registers) + the lane's offset
// (LR_PROJECT_NT_LOADS: the inputs -- read once per view -- as non-temporal loads; measured, no gain: profiles/r04_isa_project.md)
LR_DEV float lr_in(const float* p) {
#ifdef LR_PROJECT_NT_LOADS
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <bool COV3D>
LR_DEV LrGeo lr_load_geo(const float* __restrict__ wm, const float* __restrict__ ws, const float* __restrict__ wr,
                         const float* __restrict__ wc6, uint32_t ulane) {
  LrGeo g;
  const size_t lane = ulane;
  g.p[0] = lr_in(wm + 3 * lane); g.p[1] = lr_in(wm + 3 * lane + 1); g.p[2] = lr_in(wm + 3 * lane + 2);
  if (COV3D) {
    g.s[0] = wc6[6 * lane]; g.s[1] = wc6[6 * lane + 1]; g.s[2] = wc6[6 * lane + 2];
    g.q = float4{wc6[6 * lane + 3], wc6[6 * lane + 4], wc6[6 * lane + 5], 0.f};
  } else {
    g.s[0] = lr_in(ws + 3 * lane); g.s[1] = lr_in(ws + 3 * lane + 1); g.s[2] = lr_in(ws + 3 * lane + 2);
#ifdef LR_PROJECT_NT_LOADS
    const lr_f4v q = __builtin_nontemporal_load(reinterpret_cast<const lr_f4v*>(wr) + lane);
    g.q = float4{q.x, q.y, q.z, q.w};
#else
    g.q = reinterpret_cast<const float4*>(wr)[lane];
#endif
  }
  return g;
}

template <bool COV3D>
__global__ void __launch_bounds__(LR_BATCH_THREADS)
lr_project_batched_kernel(LrView v, int N, const float* __restrict__ means, const float* __restrict__ scales,
                          const float* __restrict__ rots, const float* __restrict__ opac,
                          const float* __restrict__ colors, int* __restrict__ radii, float4* __restrict__ geom,
                          uint32_t* __restrict__ ranked, uint32_t* __restrict__ big, uint32_t* __restrict__ hdr,
                          uint32_t* __restrict__ basetab, uint32_t* __restrict__ hugemask, int tile_cull, int B,
                          int S, int defer_tiles, int mid_coop, int mid_rank LR_ABLATE_PARAM) {
  // (experiment builds, LOGRAST_PROJECT_ABLATE: 1 no record stores, 2 no fill-record / radii stores, 4 no ranking atomics,
  //  8 no arithmetic -- the loop's loads and stores alone --, 16 no reservations at the end; results are garbage)
  extern __shared__ uint32_t lr_lds_ctr[];  // [S][tiles] packed (ranked | big << 16) counts, one plane per batch
  __shared__ uint32_t lr_huge_mask[LR_MAX_PLANES * LR_HUGE_WORDS];   // per batch: bit c = its 256-Gaussian chunk c deferred a rect
  __shared__ uint32_t lr_rank_dummy[64];    // where the ranking atomics of tiles that are not ranked go (see the loop)
  __shared__ uint32_t lr_mid_cnt[LR_MAX_PLANES];   // rank rows handed out per batch (rects of 5..16 tiles: common.hpp)
  const int tiles = v.gx * v.gy;
  for (int t = threadIdx.x; t < S * tiles; t += LR_BATCH_THREADS) lr_lds_ctr[t] = 0u;
  if (threadIdx.x < LR_MAX_PLANES * LR_HUGE_WORDS) lr_huge_mask[threadIdx.x] = 0u;
  if (threadIdx.x < LR_MAX_PLANES) lr_mid_cnt[threadIdx.x] = 0u;
  const uint32_t midcap = lr_mid_cap((uint32_t)B);
  uint16_t* const midrank = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(geom) + lr_midrank_off_bytes((size_t)N));
  if (blockIdx.x == 0 && threadIdx.x == 0) { hdr[LR_HDR_CULL] = tile_cull ? 1u : 0u; hdr[LR_HDR_BATCH] = (uint32_t)B; }
  __syncthreads();
  uint32_t rect_instances = 0;
  const int i_begin = blockIdx.x * (S * B), i_end = min(N, i_begin + S * B);
  uint4* const fillrec = reinterpret_cast<uint4*>(geom + LR_REC_QUADS * (size_t)N);
  const lr_cfloat* V = lr_uniform(v.view);
  const lr_cfloat* Pm = lr_uniform(v.proj);
  const int lane = (int)threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  // Every load of the loop is unconditional: a wave always reads 64 consecutive Gaussians that exist.  Only the last wave
  // of the array can be short; it reads the LAST 64 Gaussians instead ([N - 64, N): `first` slides back) and its lanes in
  // front of its own range are masked out of every store and counter like the lanes behind the end (`mine`).  Arrays of
  // fewer than 64 Gaussians: the lanes past the end repeat the last one (`ulane`).
  const uint32_t ulane = (uint32_t)min(lane, N - 1);
  const int last_first = max(N - 64, 0);
  int iw = i_begin + wave * 64;                              // the wave's first Gaussian of this iteration (scalar)
  int istep = LR_BATCH_THREADS, iend = i_end;
  if (LR_ABLATED(32)) {   // experiment builds: workgroups interleaved 1024 Gaussians at a time (wrong lists: timing only)
    iw = (int)blockIdx.x * LR_BATCH_THREADS + wave * 64; istep = (int)gridDim.x * LR_BATCH_THREADS; iend = N;
  }
  // plane of the iteration = (iw - i_begin) / B (B is a multiple of the workgroup size): counted, not divided
  int plane = 0, left_in_plane = B / LR_BATCH_THREADS;
  const float* const cov6 = v.cov3d;
  LrGeo geo;
  float nx_op, nx_c0, nx_c1, nx_c2;
  {
    const size_t i0 = (size_t)lr_sgpr((uint32_t)min(iw, last_first));
    geo = lr_load_geo<COV3D>(means + 3 * i0, scales + 3 * i0, rots + 4 * i0, COV3D ? cov6 + 6 * i0 : nullptr, ulane);
    nx_op = (opac + i0)[ulane];
    nx_c0 = (colors + 3 * i0)[3 * (size_t)ulane]; nx_c1 = (colors + 3 * i0)[3 * (size_t)ulane + 1];
    nx_c2 = (colors + 3 * i0)[3 * (size_t)ulane + 2];
  }
  // FULL waves only: every store of the loop is then unconditional.  (A store under a lane predicate sits behind an
  // `s_cbranch_execz`, and the wait-count pass cannot count instructions it may have skipped: to be sure of a load that was
  // issued BEFORE such stores it waits for "all but the unconditional instructions since" -- i.e. for the stores too, whose
  // acknowledgements then sit in every iteration's critical path.  The disassembly had `s_waitcnt vmcnt(0)` at the loop's
  // top.)  The one partial wave of the whole array is projected after the loop, by the plain per-Gaussian code.
  // (the first inputs are "used" here, i.e. waited for in front of the loop: see the wait at the loop's end)
  asm volatile("" : "+v"(geo.p[0]), "+v"(geo.p[1]), "+v"(geo.p[2]), "+v"(geo.s[0]), "+v"(geo.s[1]), "+v"(geo.s[2]),
               "+v"(geo.q.x), "+v"(geo.q.y), "+v"(geo.q.z), "+v"(geo.q.w), "+v"(nx_op), "+v"(nx_c0), "+v"(nx_c1), "+v"(nx_c2));
#ifdef LR_EXPERIMENTS
  // bits 8 + 64: memory accesses only, and every wave streams TWO blocks per iteration (its own and one half a workgroup
  // range further on), each prefetched an iteration ahead: twice the bytes in flight per wave -- is the loop bound by
  // memory-level parallelism?
  LrGeo geo2 = geo;
  float sx_op = 0.f, sx_c0 = 0.f, sx_c1 = 0.f, sx_c2 = 0.f;
  int half = 0;
  if (LR_ABLATED(64)) {
    half = ((i_end - i_begin) / 2) / LR_BATCH_THREADS * LR_BATCH_THREADS;
    iend = i_begin + half;
    const size_t j0 = (size_t)lr_sgpr((uint32_t)min(iw + half, last_first));
    geo2 = lr_load_geo<COV3D>(means + 3 * j0, scales + 3 * j0, rots + 4 * j0, COV3D ? cov6 + 6 * j0 : nullptr, ulane);
    sx_op = (opac + j0)[ulane];
    sx_c0 = (colors + 3 * j0)[3 * (size_t)ulane]; sx_c1 = (colors + 3 * j0)[3 * (size_t)ulane + 1];
    sx_c2 = (colors + 3 * j0)[3 * (size_t)ulane + 2];
  }
#endif
  for (; iw + 64 <= iend; iw += istep) {
    const size_t iws = (size_t)lr_sgpr((uint32_t)iw);        // (scalar) the Gaussian of lane 0
    if (left_in_plane == 0) { plane++; left_in_plane = B / LR_BATCH_THREADS; }
    left_in_plane--;
    uint32_t* const ctr = lr_lds_ctr + plane * tiles;
    // ---- part 1: everything that reads the raw inputs (the op sequences of lr_project_rect) ----
    LrEwa e;
    e.t[0] = lr_dot3p(V[0], V[4], V[8], geo.p[0], geo.p[1], geo.p[2], V[12]);
    e.t[1] = lr_dot3p(V[1], V[5], V[9], geo.p[0], geo.p[1], geo.p[2], V[13]);
    e.t[2] = lr_dot3p(V[2], V[6], V[10], geo.p[0], geo.p[1], geo.p[2], V[14]);
    const float tz = e.t[2];
    const float hx = lr_dot3p(Pm[0], Pm[4], Pm[8], geo.p[0], geo.p[1], geo.p[2], Pm[12]);
    const float hy = lr_dot3p(Pm[1], Pm[5], Pm[9], geo.p[0], geo.p[1], geo.p[2], Pm[13]);
    const float hw = lr_dot3p(Pm[3], Pm[7], Pm[11], geo.p[0], geo.p[1], geo.p[2], Pm[15]);
    float Sg[6];
    if (COV3D) {   // cov3D_precomp: see lr_load_inputs
      Sg[0] = geo.s[0]; Sg[1] = geo.s[1]; Sg[2] = geo.s[2]; Sg[3] = geo.q.x; Sg[4] = geo.q.y; Sg[5] = geo.q.z;
    } else {
      const float s3[3] = {geo.s[0] * v.scale_modifier, geo.s[1] * v.scale_modifier, geo.s[2] * v.scale_modifier};
      const float q4[4] = {geo.q.x, geo.q.y, geo.q.z, geo.q.w};
      float R[9];
      lr_cov3d(s3, q4, R, Sg);
    }
    // the next iteration's inputs: means / scales / rotations into the registers that just died, opacity + colour (this
    // iteration's are used at its end) into four registers of their own -- all 56 bytes one full iteration ahead
    const uint32_t olane = lr_after(ulane, e.t, hx, hy, hw, Sg);
    const float in_op = nx_op, in_c0 = nx_c0, in_c1 = nx_c1, in_c2 = nx_c2;
    {
      const size_t in = (size_t)lr_sgpr((uint32_t)min(iw + istep, last_first));   // (past the workgroup's end: loaded, never used)
      nx_op = lr_in(opac + in + olane);
      const float* __restrict__ wc = colors + 3 * in;
      nx_c0 = lr_in(wc + 3 * (size_t)olane); nx_c1 = lr_in(wc + 3 * (size_t)olane + 1); nx_c2 = lr_in(wc + 3 * (size_t)olane + 2);
      geo = lr_load_geo<COV3D>(means + 3 * in, scales + 3 * in, rots + 4 * in, COV3D ? cov6 + 6 * in : nullptr, olane);
    }
    // ---- part 2: EWA, conic, radius, rect ----
    if (LR_ABLATED(8)) {   // experiment builds: the loop's memory accesses with next to no arithmetic behind them
      float4 a0 = {e.t[0], e.t[1], hx, hy}, a1 = {hw, Sg[0], in_op, in_c0}, a2 = {in_c1, in_c2, Sg[1], Sg[2]},
             a3 = {Sg[3], Sg[4], Sg[5], tz};
      if (!LR_ABLATED(2)) {
        lr_out_store(&(radii + iws)[lane], (int)__float_as_uint(hx) & 1);
        lr_out_store(&(fillrec + iws)[lane], uint4{__float_as_uint(tz), 0xffffffffu, 0u, 0u});
      }
      if (!LR_ABLATED(1)) lr_store_records(geom + LR_REC_QUADS * iws, lane, a0, a1, a2, a3);
#ifdef LR_EXPERIMENTS
      if (LR_ABLATED(64)) {
        const size_t js = (size_t)lr_sgpr((uint32_t)min(iw + half, last_first));
        float4 b0 = {geo2.p[0], geo2.p[1], geo2.p[2], geo2.s[0]}, b1 = {geo2.s[1], geo2.s[2], geo2.q.x, geo2.q.y},
               b2 = {geo2.q.z, geo2.q.w, sx_op, sx_c0}, b3 = {sx_c1, sx_c2, 0.f, 0.f};
        const size_t jn = (size_t)lr_sgpr((uint32_t)min(iw + istep + half, last_first));
        sx_op = (opac + jn)[ulane];
        sx_c0 = (colors + 3 * jn)[3 * (size_t)ulane]; sx_c1 = (colors + 3 * jn)[3 * (size_t)ulane + 1];
        sx_c2 = (colors + 3 * jn)[3 * (size_t)ulane + 2];
        geo2 = lr_load_geo<COV3D>(means + 3 * jn, scales + 3 * jn, rots + 4 * jn, COV3D ? cov6 + 6 * jn : nullptr, ulane);
        if (!LR_ABLATED(2)) {
          lr_out_store(&(radii + js)[lane], (int)__float_as_uint(b0.x) & 1);
          lr_out_store(&(fillrec + js)[lane], uint4{__float_as_uint(b0.y), 0xffffffffu, 0u, 0u});
        }
        if (!LR_ABLATED(1)) lr_store_records(geom + LR_REC_QUADS * js, lane, b0, b1, b2, b3);
      }
#endif
      continue;
    }
    bool valid = tz > 0.2f;
    const float pw = 1.0f / (hw + 0.0000001f);
    const float nx = hx * pw, ny = hy * pw;
    if (v.ndc_cull) valid = valid && !(nx < -1.3f || nx > 1.3f || ny < -1.3f || ny > 1.3f);
    lr_ewa_t(Sg, V, v.fx, v.fy, v.tanfovx, v.tanfovy, v.filter_mode, e);
    const float det = e.a * e.c - e.b * e.b;
    valid = valid && (det != 0.0f);
    const float det_inv = 1.f / det;
    const float cA = e.c * det_inv, cB = -e.b * det_inv, cC = e.a * det_inv;
    const float rf = ceilf(lr_radius_from_cov(e.a, e.c, det));
    const float mx = ((nx + 1.0f) * (float)v.W - 1.0f) * 0.5f;
    const float my = ((ny + 1.0f) * (float)v.H - 1.0f) * 0.5f;
    valid = valid && (rf <= 1048576.f) && (fabsf(mx) < 1.0e8f) && (fabsf(my) < 1.0e8f);
    // (a culled lane converts zeros: float -> int of NaN / out-of-range values is undefined)
    const float mxs = valid ? mx : 0.f, mys = valid ? my : 0.f, rfs = valid ? rf : 0.f;
    int x0 = (int)((mxs - rfs) / 16.f), y0 = (int)((mys - rfs) / 16.f);
    int x1 = (int)(((mxs + rfs) + 15.f) / 16.f), y1 = (int)(((mys + rfs) + 15.f) / 16.f);
    x0 = min(v.gx, max(0, x0)); x1 = min(v.gx, max(0, x1));
    y0 = min(v.ty1, max(v.ty0, y0)); y1 = min(v.ty1, max(v.ty0, y1));   // [ty0, ty1) = [0, gy) unless the image is split
    const int w = x1 - x0, nt = valid ? w * (y1 - y0) : 0;
    valid = nt > 0;
    const int rad = valid ? (int)rfs : 0;
    rect_instances += (uint32_t)nt;
    // ---- part 3: support cull + ranking of the rect's tiles ----
    uint32_t slot0 = 0u, slot1 = 0u, slot2 = 0u, slot3 = 0u;
    int mrow = -1;                                             // rank row of a ranked 5..16-tile rect
    const bool small = valid && nt <= LR_RANKED_TILES;
    {
      // tile k of a rect of <= 4 tiles: one row (w >= nt), one column (w == 1) or 2x2 -- no integer division
      const bool col = w == 1, sq = (w == 2) && (nt == 4);
      const int tx1 = col ? 0 : 1, ty1 = col ? 1 : 0;
      const int tx2 = col ? 0 : (sq ? 0 : 2), ty2 = col ? 2 : (sq ? 1 : 0);
      const int tx3 = col ? 0 : (sq ? 1 : 3), ty3 = col ? 3 : (sq ? 1 : 0);
      bool k0 = true, k1 = true, k2 = true, k3 = true;
      LrSupport sup = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1};   // mode 1: every tile of the rect
      if (tile_cull) {
        sup = lr_support_prepare(mx, my, cA, cB, cC, in_op);
        const float X = (float)(x0 * LR_TILE), Y = (float)(y0 * LR_TILE);
        lr_support_tile2(sup, lr_f2{X, X + (float)(tx1 * LR_TILE)}, lr_f2{Y, Y + (float)(ty1 * LR_TILE)}, k0, k1);
        lr_support_tile2(sup, lr_f2{X + (float)(tx2 * LR_TILE), X + (float)(tx3 * LR_TILE)},
                         lr_f2{Y + (float)(ty2 * LR_TILE), Y + (float)(ty3 * LR_TILE)}, k2, k3);
        k0 = k0 || nt == 1;   // a single-tile rect holds the centre's neighbourhood: it is never dropped
      }
      const int t0 = y0 * v.gx + x0;
      const bool r0 = small && k0, r1 = small && nt > 1 && k1, r2 = small && nt > 2 && k2, r3 = small && nt > 3 && k3;
      // four returning LDS atomics in flight, all UNCONDITIONAL (under lane predicates each one sat behind a branch and
      // was followed by its own `s_waitcnt lgkmcnt(0)`: four dependent LDS round trips): a tile that is not ranked sends
      // its lane to the lane's own dummy word instead (64 words, shared by the workgroup's waves: never read)
      uint32_t* const dummy = lr_rank_dummy + lane;
      uint32_t a0 = 0u, a1 = 0u, a2 = 0u, a3 = 0u;
      if (!LR_ABLATED(4)) {
        a0 = atomicAdd(r0 ? &ctr[t0] : dummy, 1u) & 0xffffu;
        a1 = atomicAdd(r1 ? &ctr[t0 + ty1 * v.gx + tx1] : dummy, 1u) & 0xffffu;
        a2 = atomicAdd(r2 ? &ctr[t0 + ty2 * v.gx + tx2] : dummy, 1u) & 0xffffu;
        a3 = atomicAdd(r3 ? &ctr[t0 + ty3 * v.gx + tx3] : dummy, 1u) & 0xffffu;
      }
      if (small) {
        slot0 = r0 ? a0 : 0xffffffffu;
        slot1 = nt > 1 ? (r1 ? a1 : 0xffffffffu) : 0u;
        slot2 = nt > 2 ? (r2 ? a2 : 0xffffffffu) : 0u;
        slot3 = nt > 3 ? (r3 ? a3 : 0xffffffffu) : 0u;
      }
      // larger rects are only counted here (rare on uniform tiny splats; 3-4 % of a trained model's Gaussians)
      const bool huge = valid && !small && nt > defer_tiles;
      const bool mid = valid && !small && !huge;
      if (huge) {   // its 256-Gaussian chunk of the batch has work for lr_count_huge_kernel (the wave lies inside one chunk)
        const int c = (iw - (i_begin + plane * B)) >> 8;
        atomicOr(&lr_huge_mask[plane * LR_HUGE_WORDS + (c >> 5)], 1u << (c & 31));
      }
      const uint64_t midm = __builtin_amdgcn_ballot_w64(mid);
      if (midm != 0) {   // (wave-uniform)
        // Up to `mid_coop` such rects in the wave: four rects per pass, one tile per lane (lr_mid_rects; 16 lanes per rect:
        // LOGRAST_DEFER_TILES above 16 leaves the rects between to their lanes).  More -- siblings of a level-of-detail
        // tree sit in neighbouring lanes: 30 of 64 -- and every lane walks its own rect: c / 4 passes of ~170 instructions
        // against ~12 iterations of ~75 (measured on the tree-ordered C3 view: always cooperative 331 -> 426 us).
        const bool coop = mid_coop > 0 && (int)__popcll(midm) <= mid_coop;
        const bool rankable = mid && nt <= LR_MID_ROW;
        const int gxw = v.gx;
        // ranked (common.hpp): the owner takes a rank row of its batch; the counting LDS atomics return the ranks
        if (rankable && mid_rank) {
          const uint32_t r = atomicAdd(&lr_mid_cnt[plane], 1u);
          if (r < midcap) mrow = (int)r;
        }
        uint16_t* const rows = midrank + (size_t)(blockIdx.x * S + plane) * midcap * LR_MID_ROW;   // (wave-uniform)
        const bool midc = rankable && coop;
        if (coop) {
          lr_mid_rects<true>(midc, x0, y0, w, nt, sup, mrow, 0, 0, [&](int t, int ty, int tx, bool keep, int mr, int, int) {
            if (mr >= 0) {
              const uint32_t r = keep ? (atomicAdd(&ctr[ty * gxw + tx], 1u) & 0xffffu) : 0xffffu;
              rows[(size_t)mr * LR_MID_ROW + t] = (uint16_t)r;
            } else if (keep) {
              atomicAdd(&ctr[ty * gxw + tx], 0x10000u);
            }
          });
        }
        if (mid && !midc) {
          for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
              const bool keep = lr_support_tile(sup, x, y);
              if (mrow >= 0) {   // tile t = (y - y0) w + (x - x0): the order the fill reads the row in
                const uint32_t r = keep ? (atomicAdd(&ctr[y * gxw + x], 1u) & 0xffffu) : 0xffffu;
                rows[(size_t)mrow * LR_MID_ROW + (y - y0) * w + (x - x0)] = (uint16_t)r;
              } else if (keep) {
                atomicAdd(&ctr[y * gxw + x], 0x10000u);
              }
            }
        }
      }
    }
    // ---- part 4: outputs ----
    const uint32_t r0w = (uint32_t)x0 | ((uint32_t)y0 << 16), r1w = (uint32_t)x1 | ((uint32_t)y1 << 16);
    float4 g0 = float4{mx, my, cA, cB};
    float4 g1 = float4{cC, in_op, in_c0, in_c1};
    float4 g2 = float4{in_c2, tz, __uint_as_float(valid ? r0w : 0u), __uint_as_float(valid ? r1w : 0u)};
    float4 g3 = float4{__uint_as_float(slot0), __uint_as_float(slot1), __uint_as_float(slot2), __uint_as_float(slot3)};
    if (!LR_ABLATED(2)) {
      lr_out_store(&(radii + iws)[lane], rad);
      // fill record (see lr_fill_record): ranks as 16-bit halves, 0xffff = dropped by the support cull
      const uint32_t h = (uint32_t)(y1 - y0);
      uint4 fr;
      fr.x = __float_as_uint(tz);
      fr.y = !valid ? 0xffffffffu
                    : (small ? ((uint32_t)x0 | ((uint32_t)y0 << 13) | ((uint32_t)(w - 1) << 26) | ((h - 1u) << 28))
                             : ((uint32_t)x0 | ((uint32_t)y0 << 13) | (1u << 30) | (mrow >= 0 ? (1u << 31) : 0u)));
      fr.z = !valid ? 0u : (small ? ((slot0 & 0xffffu) | (slot1 << 16)) : r1w);
      fr.w = (valid && small) ? ((slot2 & 0xffffu) | (slot3 << 16)) : (mrow >= 0 ? (uint32_t)mrow : 0u);   // bit 31 of y: w = rank row
      lr_out_store(&(fillrec + iws)[lane], fr);
    }
    if (!LR_ABLATED(1)) lr_store_records(geom + LR_REC_QUADS * iws, lane, g0, g1, g2, g3);   // (scalar base + lane offset)
    // The next iteration's inputs were requested BEFORE this iteration's six stores: "at most six memory instructions
    // outstanding" says they have arrived and leaves the stores in flight.  Said here, because at the loop's top the
    // wait-count pass merges this path with the preheader's (where the same loads are the LAST instructions issued) into
    // `s_waitcnt vmcnt(0)` -- every store of the iteration acknowledged before the next one starts.  (An explicit wait can
    // only be too weak, never wrong: the pass still adds whatever a use needs.)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(LR_WAIT_VMCNT(6));
    __builtin_amdgcn_sched_barrier(0);
  }
  if (iw < i_end && !LR_ABLATED(32)) {
    // the partial wave (at most one in the whole grid: the end of the array) -- lr_project_one, the code of the other
    // projection kernels: the same op sequences, so the same records bit for bit
    const int i = iw + lane;
    if (i < i_end) {
      const int pl = (iw - i_begin) / B;
      const LrLdsCounters lc{lr_lds_ctr + pl * tiles};
      const LrInputs in = lr_load_inputs(i, means, scales, rots, opac, colors, v.cov3d);
      float4 g0, g1, g2, g3;
      int rad;
      bool huge;
      lr_project_one<true>(v, in, tile_cull, lc, g0, g1, g2, g3, rad, rect_instances, huge, defer_tiles);
      if (huge) {
        const int c = (i - (i_begin + pl * B)) >> 8;
        atomicOr(&lr_huge_mask[pl * LR_HUGE_WORDS + (c >> 5)], 1u << (c & 31));
      }
      radii[i] = rad;
      fillrec[i] = lr_fill_record(g2, g3, rad);
      geom[LR_REC_QUADS * (size_t)i + 0] = g0; geom[LR_REC_QUADS * (size_t)i + 1] = g1;
      geom[LR_REC_QUADS * (size_t)i + 2] = g2; geom[LR_REC_QUADS * (size_t)i + 3] = g3;
    }
  }
  __syncthreads();
  const int nplanes = min(S, (i_end - i_begin + B - 1) / B);   // batches this workgroup really holds
  if (!LR_ABLATED(16))
    lr_reserve_batches(lr_lds_ctr, tiles, 0, tiles, nplanes, tiles, ranked, big, basetab + (size_t)blockIdx.x * S * tiles);
  if ((int)threadIdx.x < nplanes * LR_HUGE_WORDS) {   // complete: the barrier after the Gaussian loop
    const uint32_t mw = lr_huge_mask[threadIdx.x];
    hugemask[(size_t)blockIdx.x * S * LR_HUGE_WORDS + threadIdx.x] = mw;   // (every word of every batch is written: nothing to clear)
    if (mw) atomicOr(&hdr[LR_HDR_HUGE], 1u);
  }
  lr_commit_rect_count<LR_BATCH_THREADS / 64>(rect_instances, hdr);
}

// ---- band views ---------------------------------------------------------------------------------------------------
// A view that owns a band of tile rows ([ty0, ty1) a proper part of the grid: one rank's share of an image split across
// GPUs) is handed ALL Gaussians and keeps few: most end with an empty rect.  Returning early from the loop
// above saves nothing -- in Gaussian order nearly every wave holds a survivor and runs the whole projection (100 M
// Gaussians at 3840x2160, 19.5 M in a band of 17 of the 135 tile rows: 3.3 ms either way).  This kernel works in two
// phases per wave:
//   A  every Gaussian: its rect from the 40 bytes of means3D / scales / rotations (lr_project_rect, ~300 instructions)
//      and radii[]; a survivor's index and those ten floats go to a per-wave ring in LDS;
//   B  whenever the ring holds 64: one FULL wave of survivors runs the rest of the projection (support test, ranking,
//      record; ~1000 instructions).  Its inputs come from the ring -- gathering them again from memory made the texture
//      addresser the bound (64 distinct lines per load instruction) -- except opacity and colour, which only survivors
//      ever cost and which are requested one firing ahead of their use.
// A Gaussian without a rect costs 44 bytes and neither a record nor a fill record.  The survivors' fill records go,
// compacted per workgroup, to the front of the workgroup's own range of the fill-record array, their indices to the same
// slots of an index array behind it (survcount[w] of them in workgroup w's range): lr_fill_kernel / lr_count_huge_kernel
// walk those slots -- coalesced, and four fifths of their workgroups return after one read.  Ranks inside a (batch, tile) run are handed out
// in ring order, not Gaussian order -- any order will do, the lists are sorted by (depth, id) afterwards.
// LDS: S counter planes over the BAND's tiles only (counters, reservations and slot-table entries of other tiles are
// neither cleared nor written nor read: a band of 17 rows at 3840x2160 is 4080 of 32400 tiles) + 88 KB of rings.
// Measured (MI355X, 100 M Gaussians, 19.5 M in the band): projection 3.13 -> 1.88 ms (phase A alone 1.21, of which the
// 44 bytes per Gaussian are 1.00), fill 0.89 -> 0.62, slot-table rebase 0.25 -> 0.04.
#define LR_RING 128                       // entries per wave: < 64 pending before a phase-A step, <= 64 added by it
#define LR_RING_FLOATS 10                 // mean 3, scale 3, quaternion 4 (or the six covariance floats)
#define LR_BAND_STATIC_LDS ((LR_BATCH_THREADS / 64) * LR_RING * (LR_RING_FLOATS + 1) * 4 + 128)   // rings + the small per-workgroup words
struct LrLdsBandCounters {                // LrLdsCounters over tiles [lo, ...)
  uint32_t* ctr;
  int lo;
  LR_DEV uint32_t rank(int tile) const { return atomicAdd(&ctr[tile - lo], 1u) & 0xffffu; }
  LR_DEV void count_big(int tile) const { atomicAdd(&ctr[tile - lo], 0x10000u); }
};
__global__ void __launch_bounds__(LR_BATCH_THREADS)
lr_project_band_kernel(LrView v, int N, const float* __restrict__ means, const float* __restrict__ scales,
                       const float* __restrict__ rots, const float* __restrict__ opac,
                       const float* __restrict__ colors, int* __restrict__ radii, float4* __restrict__ geom,
                       uint32_t* __restrict__ ranked, uint32_t* __restrict__ big, uint32_t* __restrict__ hdr,
                       uint32_t* __restrict__ basetab, uint32_t* __restrict__ hugemask,
                       uint32_t* __restrict__ survcount, int tile_cull, int B, int S, int defer_tiles LR_ABLATE_PARAM) {
  extern __shared__ uint32_t lr_lds_ctr[];  // [S][band tiles] packed (ranked | big << 16) counts, one plane per batch
  // which 256-SLOT chunks of the workgroup's fill-record range hold a deferred rect (lr_count_huge_kernel walks slots):
  // the words of "batch" p = slots [p B, (p + 1) B) of the range
  __shared__ uint32_t lr_huge_mask[LR_MAX_PLANES * LR_HUGE_WORDS];
  __shared__ uint32_t lr_slot_cursor;       // survivors of this workgroup so far = its next free fill-record slot
  __shared__ uint32_t lr_ring_idx[LR_BATCH_THREADS / 64][LR_RING];
  __shared__ float lr_ring_in[LR_BATCH_THREADS / 64][LR_RING_FLOATS][LR_RING];
  const int tiles = v.gx * v.gy;
  const int t_lo = v.ty0 * v.gx, t_hi = v.ty1 * v.gx, band_tiles = t_hi - t_lo;
  for (int t = threadIdx.x; t < S * band_tiles; t += LR_BATCH_THREADS) lr_lds_ctr[t] = 0u;
  if (threadIdx.x < LR_MAX_PLANES * LR_HUGE_WORDS) lr_huge_mask[threadIdx.x] = 0u;
  if (threadIdx.x == 0) lr_slot_cursor = 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    hdr[LR_HDR_CULL] = tile_cull ? 1u : 0u; hdr[LR_HDR_BATCH] = (uint32_t)B; hdr[LR_HDR_SPARSE] = 1u;
    hdr[LR_HDR_SPAN] = (uint32_t)(S * B);
  }
  __syncthreads();
  uint32_t rect_instances = 0;
  const int i_begin = blockIdx.x * (S * B), i_end = min(N, i_begin + S * B);
  uint4* const fillrec = reinterpret_cast<uint4*>(geom + LR_REC_QUADS * (size_t)N);    // compacted: slot, not Gaussian
  uint32_t* const survivor = reinterpret_cast<uint32_t*>(fillrec + N);                 // Gaussian index of a slot
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  uint32_t* const ring_idx = lr_ring_idx[wave];
  float (*const ring_in)[LR_RING] = lr_ring_in[wave];
  int i = i_begin + (int)threadIdx.x;
  LrInputs nxt;
  if (i < i_end) nxt = lr_load_inputs<true>(i, means, scales, rots, opac, colors, v.cov3d);
  int head = 0, npend = 0;                                   // wave-uniform: ring entries [head, head + npend)
  int pj = -1;                                               // the batch fired last time: its opacity / colour are in flight
  bool have = false;                                         // wave-uniform
  LrInputs pin = nxt;                                        // (any value: not read while pj < 0)
  // One loop for both phases and for the drain at the end, so that phase B -- the rest of the projection, inlined --
  // exists at ONE call site (with three it became a real call with its arguments in scratch memory: 2.5 -> 5.2 ms).
  for (;;) {
    const bool more = (i & ~63) < i_end;                     // wave-uniform
    if (more) {
      const bool mine = i < i_end;
      const LrInputs in = nxt;
      if (i + LR_BATCH_THREADS < i_end) nxt = lr_load_inputs<true>(i + LR_BATCH_THREADS, means, scales, rots, opac, colors, v.cov3d);
      int rad = 0;
      if (mine) {
        LrRect rc;
        if (LR_ABLATED(2)) rad = (in.p[0] + in.s[1] + in.q.z == 1.2345e30f) ? 1 : 0;   // experiment builds: loads only
        else if (lr_project_rect(v, in, rc)) rad = rc.rad;
        if (LR_ABLATED(1)) rad = 0;                                                      // experiment builds: no phase B
        radii[i] = rad;
      }
      const uint64_t rect_mask = __ballot(rad > 0);
      if (rad > 0) {
        const int e = (head + npend + __popcll(rect_mask & ((1ull << lane) - 1ull))) & (LR_RING - 1);
        ring_idx[e] = (uint32_t)i;
        ring_in[0][e] = in.p[0]; ring_in[1][e] = in.p[1]; ring_in[2][e] = in.p[2];
        ring_in[3][e] = in.s[0]; ring_in[4][e] = in.s[1]; ring_in[5][e] = in.s[2];
        ring_in[6][e] = in.q.x; ring_in[7][e] = in.q.y; ring_in[8][e] = in.q.z; ring_in[9][e] = in.q.w;
      }
      npend += __popcll(rect_mask);
      i += LR_BATCH_THREADS;
    }
    // (the ring is private to the wave: LDS executes a wave's accesses in program order and the compiler keeps the order
    // of may-alias LDS accesses -- no fence: an acquire fence here turns every later read of the view's matrices into a
    // vector load behind s_waitcnt vmcnt(0), i.e. behind the prefetch of the next iteration)
    const bool fire = more ? npend >= 64 : (npend > 0 || have);
    if (!more && !fire) break;
    if (!fire) continue;
    // take (up to) 64 survivors off the ring and request their opacity / colour; run the batch taken last time
    int j = -1;
    LrInputs nin = pin;
    const bool took = npend > 0;
    if (took) {
      __builtin_amdgcn_wave_barrier();
      if (lane < npend) {
        const int e = (head + lane) & (LR_RING - 1);
        j = (int)ring_idx[e];
        nin.p[0] = ring_in[0][e]; nin.p[1] = ring_in[1][e]; nin.p[2] = ring_in[2][e];
        nin.s[0] = ring_in[3][e]; nin.s[1] = ring_in[4][e]; nin.s[2] = ring_in[5][e];
        nin.q = float4{ring_in[6][e], ring_in[7][e], ring_in[8][e], ring_in[9][e]};
        nin.op = opac[j];
        nin.c[0] = colors[3 * j]; nin.c[1] = colors[3 * j + 1]; nin.c[2] = colors[3 * j + 2];
      }
      const int n_taken = min(npend, 64);
      head = (head + n_taken) & (LR_RING - 1);
      npend -= n_taken;
    }
    if (have) {
      // phase B for Gaussian pj (< 0: idle lane) with inputs pin, run by the whole wave: the records leave through the
      // same 4x4 transpose as in the full-view loop, so that a store instruction writes complete 64-byte lines
      float4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, g2 = g0, g3 = g0;
      int rad = 0;
      bool huge = false;
      if (pj >= 0) {
        const int dj = pj - i_begin;                           // (at most LR_MAX_PLANES batches per workgroup: no division)
        const int plane = (dj >= B) + (dj >= 2 * B) + (dj >= 3 * B);
        const LrLdsBandCounters ctr{lr_lds_ctr + plane * band_tiles, t_lo};
        lr_project_one<true>(v, pin, tile_cull, ctr, g0, g1, g2, g3, rad, rect_instances, huge, defer_tiles);
      }
      // slots: the workgroup compacts into its own range of the fill-record array (one counter for all workgroups was a
      // same-address memory-side atomic per firing: +1.2 ms at 100 M Gaussians)
      const uint64_t live = __ballot(pj >= 0);               // a prefix of the lanes
      uint32_t slot0 = 0u;
      if (lane == 0) slot0 = atomicAdd(&lr_slot_cursor, (uint32_t)__popcll(live));
      slot0 = (uint32_t)i_begin + (uint32_t)lr_readlane_i((int)slot0, 0);
      if (pj >= 0) {                                         // 1 KB + 256 B of contiguous stores per wave
        fillrec[slot0 + (uint32_t)lane] = lr_fill_record(g2, g3, rad);
        survivor[slot0 + (uint32_t)lane] = (uint32_t)pj;
        if (huge) {   // the chunk of its SLOT has work for lr_count_huge_kernel
          const int ds = (int)(slot0 + (uint32_t)lane) - i_begin;
          const int sp = (ds >= B) + (ds >= 2 * B) + (ds >= 3 * B);
          const int c = (ds - sp * B) >> 8;
          atomicOr(&lr_huge_mask[sp * LR_HUGE_WORDS + (c >> 5)], 1u << (c & 31));
        }
      }
      const int m = lane & 3;
      lr_quad_transpose(g0, g1, g2, g3, m);                  // g<k> = quad m of the record of lane (lane - m + k)
      const int j0 = __builtin_amdgcn_update_dpp(0, pj, 0x00, 0xf, 0xf, true), j1 = __builtin_amdgcn_update_dpp(0, pj, 0x55, 0xf, 0xf, true),
                j2 = __builtin_amdgcn_update_dpp(0, pj, 0xAA, 0xf, 0xf, true), j3 = __builtin_amdgcn_update_dpp(0, pj, 0xFF, 0xf, 0xf, true);
      if (j0 >= 0) geom[LR_REC_QUADS * (size_t)j0 + m] = g0;
      if (j1 >= 0) geom[LR_REC_QUADS * (size_t)j1 + m] = g1;
      if (j2 >= 0) geom[LR_REC_QUADS * (size_t)j2 + m] = g2;
      if (j3 >= 0) geom[LR_REC_QUADS * (size_t)j3 + m] = g3;
    }
    pj = j; pin = nin; have = took;
  }
  __syncthreads();
  const int nplanes = min(S, (i_end - i_begin + B - 1) / B);   // batches this workgroup really holds
  lr_reserve_batches(lr_lds_ctr, band_tiles, t_lo, t_hi, nplanes, tiles, ranked, big, basetab + (size_t)blockIdx.x * S * tiles);
  if (threadIdx.x == 0) survcount[blockIdx.x] = lr_slot_cursor;
  if ((int)threadIdx.x < nplanes * LR_HUGE_WORDS) {
    const uint32_t mw = lr_huge_mask[threadIdx.x];
    hugemask[(size_t)blockIdx.x * S * LR_HUGE_WORDS + threadIdx.x] = mw;
    if (mw) atomicOr(&hdr[LR_HDR_HUGE], 1u);
  }
  lr_commit_rect_count<LR_BATCH_THREADS / 64>(rect_instances, hdr);
}

// Rects of more than `defer_tiles` tiles, left out by lr_project_batched_kernel: one wave per rect, lanes = tiles, so
// the support tests of a rect run 64 at a time and the LDS atomics never collide; counts go through per-workgroup
// LDS counters (one memory-side atomic per touched tile and workgroup).  A workgroup walks chunks of `chunk` Gaussians and
// skips those whose bit of the batch's 128-bit mask is clear (hugemask: one bit per 256 Gaussians; the common case --
// small splats only -- returns after one read of the header flag); a chunk with a handful of deferred rects counts them
// straight into the global counters.
#define LR_HUGE_DIRECT 8   // deferred rects per 256-Gaussian chunk up to which they are counted without the LDS plane
__global__ void __launch_bounds__(256)
lr_count_huge_kernel(int N, int gx, int tiles, const float4* __restrict__ geom, uint32_t* __restrict__ big,
                     const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ hugemask, int B, int tile_cull,
                     int defer_tiles, int chunk) {
  const int n_all = N;                                       // (the arrays behind the records are laid out for all N)
  extern __shared__ uint32_t lr_lds_ctr[];                   // [tiles] counters | [256] chunks with work | their number
  uint32_t* const lr_chunk_todo = lr_lds_ctr + tiles;
  if (!hdr[LR_HDR_HUGE]) return;                             // no workgroup deferred anything: the common case
  // band views (lr_project_band_kernel): projection workgroup w left the fill records of its survcount[w] survivors in
  // slots [w * span, ...), their Gaussian indices in the same slots of the array behind the fill records
  const bool sparse = hdr[LR_HDR_SPARSE] != 0u;
  const int span = sparse ? (int)hdr[LR_HDR_SPAN] : 1;       // slots per projection workgroup (a multiple of B and of chunk)
  const uint32_t* __restrict__ survcount = hugemask + lr_hugemask_words((uint32_t)((N + B - 1) / B));
  // (the grid is capped at a few workgroups per CU: each walks its share of the chunks and skips those whose batches
  // deferred nothing -- 58 K workgroups that only return cost a 30 M-Gaussian view 30 us.  The flags of up to 256 chunks
  // are fetched by as many threads at once: one dependent scalar read per chunk was a 28 us chain of round trips.)
  for (int first = blockIdx.x; (long long)first * chunk < N; first += 256 * gridDim.x) {
  {
    const long long cid = (long long)first + (long long)threadIdx.x * gridDim.x;
    uint32_t any = 0;
    if (cid * chunk < N) {
      // the chunk's 256-Gaussian (band views: 256-slot) pieces: bit c of batch b's mask (lr_project_batched_kernel /
      // lr_project_band_kernel); a piece lies inside one batch (batches are multiples of 1024)
      const int base = (int)cid * chunk;
      for (int j = base; j < min(N, base + chunk); j += 256) {
        const int b = j / B, c = (j - b * B) >> 8;
        any |= (hugemask[(size_t)b * LR_HUGE_WORDS + (c >> 5)] >> (c & 31)) & 1u;
      }
    }
    __syncthreads();                                         // (the previous round's readers are done)
    if (threadIdx.x == 0) lr_chunk_todo[256] = 0u;
    __syncthreads();
    if (any) lr_chunk_todo[atomicAdd(&lr_chunk_todo[256], 1u)] = threadIdx.x;   // compact list of the chunks with work
    __syncthreads();
  }
  const int ntodo = (int)lr_chunk_todo[256];
  const int lane = threadIdx.x & 63;
  const uint4* __restrict__ fill = reinterpret_cast<const uint4*>(geom + LR_REC_QUADS * (size_t)n_all);
  const uint32_t* __restrict__ survivor = reinterpret_cast<const uint32_t*>(fill + n_all);
  for (int q = 0; q < ntodo; q++) {
  const int chunk_id = first + (int)lr_chunk_todo[q] * (int)gridDim.x;
  const int base = chunk_id * chunk;
  int x0 = 0, y0 = 0, w = 0, nt = 0;
  LrSupport sup = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1};
  auto load = [&](int i) {   // this thread's Gaussian (band views: slot) of the chunk: its rect, if it was deferred
    x0 = 0; y0 = 0; w = 0; nt = 0;
    if (i < N && (!sparse || i - (i / span) * span < (int)survcount[i / span])) {
      const uint4 fr = fill[i];
      const size_t gid = sparse ? (size_t)survivor[i] : (size_t)i;
      if (fr.y != 0xffffffffu && (fr.y & (1u << 30))) {
        x0 = (int)(fr.y & 0x1fffu); y0 = (int)((fr.y >> 13) & 0x1fffu);
        w = (int)(fr.z & 0xffffu) - x0;
        nt = w * ((int)(fr.z >> 16) - y0);
        if (nt > defer_tiles && tile_cull) {
          const float4 g0 = geom[LR_REC_QUADS * gid + 0], g1 = geom[LR_REC_QUADS * gid + 1];
          sup = lr_support_prepare(g0.x, g0.y, g0.z, g0.w, g1.x, g1.y);
        }
      }
    }
  };
  load(base + (int)threadIdx.x);
  // A chunk with a handful of deferred rects (a trained model's few large splats, spread evenly: one or two per chunk
  // with work) counts them straight into the dense global counters -- clearing and flushing the workgroup's 8160 LDS
  // counters for ~40 instances was most of this kernel's 53 us there.  (The barrier also separates the previous chunk's
  // flush from this chunk's clear.)
  const bool direct = (chunk == 256) && __syncthreads_count(nt > defer_tiles) <= LR_HUGE_DIRECT;
  if (!direct) {
    if (chunk != 256) __syncthreads();
    for (int t = threadIdx.x; t < tiles; t += 256) lr_lds_ctr[t] = 0u;
    __syncthreads();
  }
  for (int k = 0; k < chunk / 256; k++) {
    if (k > 0) load(base + k * 256 + (int)threadIdx.x);
    uint64_t m = __ballot(nt > defer_tiles);
    while (m) {
      const int src = __builtin_ctzll(m);
      m &= m - 1;
      const int bx0 = lr_readlane_i(x0, src), by0 = lr_readlane_i(y0, src), bw = lr_readlane_i(w, src),
                bn = lr_readlane_i(nt, src);
      LrSupport bs;
      bs.mx = lr_readlane_f(sup.mx, src); bs.my = lr_readlane_f(sup.my, src);
      bs.A = lr_readlane_f(sup.A, src); bs.B = lr_readlane_f(sup.B, src); bs.C = lr_readlane_f(sup.C, src);
      bs.tau = lr_readlane_f(sup.tau, src); bs.ex = lr_readlane_f(sup.ex, src); bs.ey = lr_readlane_f(sup.ey, src);
      bs.iA = lr_readlane_f(sup.iA, src); bs.iC = lr_readlane_f(sup.iC, src);
      bs.mode = lr_readlane_i(sup.mode, src);
      for (int t = lane; t < bn; t += 64) {
        const int ty = t / bw, tx = t - ty * bw;
        if (lr_support_tile(bs, bx0 + tx, by0 + ty)) {
          const int tile = (by0 + ty) * gx + (bx0 + tx);
          if (direct) atomicAdd(&big[tile], 1u); else atomicAdd(&lr_lds_ctr[tile], 1u);
        }
      }
    }
  }
  if (!direct) {
    __syncthreads();
    for (int t = threadIdx.x; t < tiles; t += 256) {
      const uint32_t c = lr_lds_ctr[t];
      if (c) atomicAdd(&big[t], c);   // dense counters in batched mode
    }
  }
  }
  }
}

// Does the batched projection of this view run in its band form (lr_project_batched_kernel<true>)?
// Does the batched projection of this view run in its band form (lr_project_band_kernel)?  A proper band of tile rows
// whose counter plane fits beside the rings (a band of more than half of a 3840x2160 grid does not: the full-view kernel).
#define LR_BAND_LDS_BYTES (160 * 1024 - 512)
bool lr_band_sparse(const LrView& v, int batch) {
  LR_KNOB(sparse_knob, "LOGRAST_BAND_SPARSE", 1);
  const size_t plane = sizeof(uint32_t) * (size_t)((v.ty1 - v.ty0) * v.gx);
  return sparse_knob && batch > 0 && (v.ty0 > 0 || v.ty1 < v.gy) && v.ty1 > v.ty0 &&
         plane + LR_BAND_STATIC_LDS <= LR_BAND_LDS_BYTES;
}

void lr_launch_project(const LrView& v, int N, const float* means, const float* scales, const float* rots,
                       const float* opac, const float* colors, int* radii, void* geom, uint32_t* ranked,
                       uint32_t* big, uint32_t* hdr, uint32_t* basetab, int batch, int planes, int tile_cull,
                       hipStream_t s) {
  if (N <= 0) return;
  lr_prof_begin(LRK_PROJECT, s);
  if (batch > 0) {
    const int tiles = v.gx * v.gy;
    const size_t lds = sizeof(uint32_t) * (size_t)tiles;
    static bool attr_set = false;
    if (!attr_set) {  // > 64 KB of dynamic LDS needs the opt-in
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lr_project_batched_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LR_BATCH_LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lr_project_batched_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LR_BATCH_LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lr_project_band_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LR_BAND_LDS_BYTES - LR_BAND_STATIC_LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lr_count_huge_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LR_BATCH_LDS_BYTES);
      attr_set = true;
    }
    LR_KNOB(defer_tiles, "LOGRAST_DEFER_TILES", LR_COOP_TILES);
    LR_KNOB(chunk_k, "LOGRAST_HUGE_CHUNK", LR_HUGE_CHUNK);
    const int chunk = chunk_k >= 256 ? chunk_k / 256 * 256 : 256;
    const int batches = (N + batch - 1) / batch;
    const int groups = (batches + planes - 1) / planes;     // workgroups: `planes` consecutive batches each
    uint32_t* hugecount = basetab + (size_t)batches * tiles;   // (hugemask[batches][LR_HUGE_WORDS]: common.hpp)
    LR_KNOB(mid_coop, "LOGRAST_MID_COOP", 16);
    LR_KNOB(mid_rank, "LOGRAST_MID_RANK", 1);
#ifdef LR_EXPERIMENTS
    static const int ablate = lr_env_int("LOGRAST_PROJECT_ABLATE", 0);   // timing experiments (tools/): see the band kernel
#endif
    if (lr_band_sparse(v, batch)) {
      // planes over the band's tiles only: as many batches per workgroup as fit beside the rings (at most 4)
      const int band_tiles = (v.ty1 - v.ty0) * v.gx;
      int bp = (int)((LR_BAND_LDS_BYTES - LR_BAND_STATIC_LDS) / (sizeof(uint32_t) * (size_t)band_tiles));
      bp = bp > LR_MAX_PLANES ? LR_MAX_PLANES : bp;
      const int bgroups = (batches + bp - 1) / bp;
      hipLaunchKernelGGL(lr_project_band_kernel, dim3(bgroups), dim3(LR_BATCH_THREADS),
                         sizeof(uint32_t) * (size_t)band_tiles * bp, s, v, N, means, scales, rots, opac, colors, radii,
                         reinterpret_cast<float4*>(geom), ranked, big, hdr, basetab, hugecount,
                         hugecount + lr_hugemask_words((uint32_t)batches), tile_cull, batch, bp, defer_tiles LR_ABLATE_PASS(ablate));
    } else {
      if (v.cov3d)
        hipLaunchKernelGGL(lr_project_batched_kernel<true>, dim3(groups), dim3(LR_BATCH_THREADS), lds * planes, s, v, N,
                           means, scales, rots, opac, colors, radii, reinterpret_cast<float4*>(geom), ranked, big, hdr,
                           basetab, hugecount, tile_cull, batch, planes, defer_tiles, mid_coop, mid_rank LR_ABLATE_PASS(ablate));
      else
        hipLaunchKernelGGL(lr_project_batched_kernel<false>, dim3(groups), dim3(LR_BATCH_THREADS), lds * planes, s, v, N,
                           means, scales, rots, opac, colors, radii, reinterpret_cast<float4*>(geom), ranked, big, hdr,
                           basetab, hugecount, tile_cull, batch, planes, defer_tiles, mid_coop, mid_rank LR_ABLATE_PASS(ablate));
    }
    lr_prof_end(LRK_PROJECT, s);
    lr_prof_begin(LRK_RESERVED, s);
    hipLaunchKernelGGL(lr_count_huge_kernel, dim3(min((N + chunk - 1) / chunk, 2048)), dim3(256), lds + 1028, s, N, v.gx,
                       tiles, reinterpret_cast<const float4*>(geom), big, (const uint32_t*)hdr, (const uint32_t*)hugecount,
                       batch, tile_cull, defer_tiles, chunk);
    lr_prof_end(LRK_RESERVED, s);
    return;
  } else {
    LR_KNOB(max_blocks, "LOGRAST_PROJECT_BLOCKS", 512);  // 2 workgroups per CU: measured optimum
    int blocks = (N + 255) / 256;
    if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL(lr_project_kernel, dim3(blocks), dim3(256), 0, s, v, N, means, scales, rots, opac, colors, radii,
                       reinterpret_cast<float4*>(geom), ranked, big, hdr, tile_cull);
  }
  lr_prof_end(LRK_PROJECT, s);
}

// ---- tile scan --------------------------------------------------------------------------------------
// One 1024-thread workgroup: exclusive scan of the per-tile counts into offsets[T+1], fill cursors, header, and
// the longest-first dispatch order.  T is 8160 at 1080p / 32400 at 4K, so one workgroup is enough -- but then the
// kernel is a pure latency chain, so every per-tile value is loaded ONCE (independent loads, one 64-byte counter
// line each) and kept in registers: thread t owns tiles [t*CH, (t+1)*CH), CH = ceil(T/1024) <= CHMAX.
template <int CHMAX>
__global__ void __launch_bounds__(1024)
lr_scan_kernel(uint32_t* __restrict__ state, uint32_t tiles, uint32_t cs, uint32_t big_off) {
  __shared__ uint32_t part[64];
  __shared__ uint32_t hist[256];
  const uint32_t* ranked = state + lr_ranked_off(tiles);
  const uint32_t* big = state + big_off;
  uint32_t* offsets = state + lr_offsets_off(tiles);
  uint32_t* cursor = state + lr_cursor_off(tiles);
  uint32_t* order = state + lr_order_off(tiles);
  const uint32_t tid = threadIdx.x;
  uint32_t lmax = 0;
  const uint32_t chunk = (tiles + 1023u) / 1024u;
  const uint32_t b = tid * chunk;
  if (tid < 256) hist[tid] = 0u;
  uint32_t nr[CHMAX], tot[CHMAX];
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < CHMAX; k++) {
    const uint32_t t = b + (uint32_t)k;
    const bool in = (uint32_t)k < chunk && t < tiles;
    nr[k] = in ? ranked[t * cs] : 0u;
    tot[k] = nr[k] + (in ? big[t * cs] : 0u);
    sum += tot[k];
  }
  // inclusive scan of the 1024 per-thread sums: shuffle scan inside each wave, then the 16 wave totals
  uint32_t inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(inc, d);
    if ((int)(tid & 63u) >= d) inc += up;
  }
  if ((tid & 63u) == 63u) part[tid >> 6] = inc;
  __syncthreads();  // also publishes the zeroed hist[]
  if (tid < 16u) {
    uint32_t w = part[tid];
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const uint32_t up = __shfl_up(w, d);
      if ((int)tid >= d) w += up;
    }
    part[32 + tid] = w;  // inclusive wave totals
  }
  __syncthreads();
  inc += (tid >> 6) ? part[32 + (tid >> 6) - 1] : 0u;
  const uint32_t grand = part[32 + 15];
  const uint32_t run0 = inc - sum;  // exclusive prefix of this thread's chunk
  // Where the per-tile stores go (measured): for the wide variants (4K / 8K grids, 32 / 128 tiles per thread) after
  // the last barrier -- a barrier waits for the workgroup's outstanding global stores (4K: 174 -> 108 us) --, for
  // 1080p here, where they overlap the LDS phases below (25 vs 28 us).
  constexpr bool LATE = CHMAX > 8;
  uint32_t run = run0;
  // Length buckets scaled to the longest list of this view (a fixed 16 keys per bucket put every list above 4080 keys
  // -- at 30 M Gaussians all of them -- into the last bucket: no order among them, and 8160 LDS atomics on one word).
#pragma unroll
  for (int k = 0; k < CHMAX; k++) lmax = max(lmax, tot[k]);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, d));
  if ((tid & 63u) == 0u) part[16 + (tid >> 6)] = lmax;      // (part[0..15] / part[32..47] hold the scan's wave totals)
  __syncthreads();
  {
    uint32_t m = part[16 + (tid & 15u)];
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
    lmax = m;                                                // longest list of the view, in every thread
  }
  // bucket(L) = L / 32 below 4096 keys (128 buckets: the sort's size classes start at multiples of 32, so every tile in
  // front of a list of more than 1024 keys in order[] holds at least 1024 itself -- lr_launch_sort sizes its grids by
  // that), 128 + (L - 4096) >> lsh above, lsh from the longest list so that the last bucket is 255
  const uint32_t over = lmax > 4096u ? lmax - 4096u : 0u;
  const uint32_t lsh = over > 127u ? (uint32_t)(25 - __clz((int)over)) : 0u;    // over >> lsh <= 127
  auto len_bucket = [&](uint32_t len) -> uint32_t { return len < 4096u ? len >> 5 : 128u + ((len - 4096u) >> lsh); };
#pragma unroll
  for (int k = 0; k < CHMAX; k++) {
    const uint32_t t = b + (uint32_t)k;
    if ((uint32_t)k < chunk && t < tiles) {
      atomicAdd(&hist[len_bucket(tot[k])], 1u);
      if (!LATE) {
        offsets[t] = run;
        cursor[t * LR_CTR_STRIDE] = run + nr[k];  // big instances go behind the ranked ones (64 B apart: the fill's atomics hit random tiles)
        run += tot[k];
      }
    }
  }
  // Longest-processing-time-first dispatch order for the blend kernels: a tile's list is walked serially by
  // its waves, so the longest lists must start first or they become the tail of the launch.  Counting sort
  // of the tiles into 256 length buckets (16 entries wide), longest bucket first.
  __syncthreads();
  if (tid < 64u) {  // one wave turns the 256 bucket counts into exclusive starts, longest bucket first
    uint32_t c[4], tt = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { c[i] = hist[255 - (4 * (int)tid + i)]; tt += c[i]; }
    uint32_t incw = tt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(incw, d);
      if ((int)tid >= d) incw += up;
    }
    uint32_t start = incw - tt;
#pragma unroll
    for (int i = 0; i < 4; i++) { hist[255 - (4 * (int)tid + i)] = start; start += c[i]; }
  }
  __syncthreads();
  run = run0;
#pragma unroll
  for (int k = 0; k < CHMAX; k++) {
    const uint32_t t = b + (uint32_t)k;
    const bool in = (uint32_t)k < chunk && t < tiles;
    if (in) {
      order[atomicAdd(&hist[len_bucket(tot[k])], 1u)] = t;
      if (LATE) {
        offsets[t] = run;
        cursor[t * LR_CTR_STRIDE] = run + nr[k];
        run += tot[k];
      }
    }
  }
  if (tid == 0u && lmax) atomicMax(&state[LR_HDR_MAXLEN], lmax);
  if (tid == 1023) {
    offsets[tiles] = grand;
    state[LR_HDR_NUM] = grand;
    state[LR_HDR_OVERFLOW] = 0u;
  }
}

// cs = word stride of the ranked / big counters.  The unbatched projection spreads them 64 B apart (one returning
// atomic per INSTANCE on random tiles: dense counters share lines and halve the atomic rate); the batched projection
// reserves per (batch, tile) with consecutive threads on consecutive tiles, where dense counters are what lets the
// memory side merge a wave's 64 atomics into 4 line operations (project 104 -> 63 us on C2, and this kernel reads
// 16x fewer lines).
void lr_launch_scan(uint32_t* state, uint32_t tiles, uint32_t cs, uint32_t big_off, hipStream_t s) {
  lr_prof_begin(LRK_SCAN, s);
  const uint32_t chunk = (tiles + 1023u) / 1024u;
  if (chunk <= 8)
    hipLaunchKernelGGL(lr_scan_kernel<8>, dim3(1), dim3(1024), 0, s, state, tiles, cs, big_off);
  else if (chunk <= 32)
    hipLaunchKernelGGL(lr_scan_kernel<32>, dim3(1), dim3(1024), 0, s, state, tiles, cs, big_off);
  else
    hipLaunchKernelGGL(lr_scan_kernel<128>, dim3(1), dim3(1024), 0, s, state, tiles, cs, big_off);  // up to 131072 tiles (8K x 4K)
  lr_prof_end(LRK_SCAN, s);
}

// ---- batched mode: absolute slots ---------------------------------------------------------------------------
// basetab[b][t] (start of batch b's run inside tile t's list) + offsets[t] (start of the tile's list), once per
// (batch, tile) here instead of once per INSTANCE in the fill kernel, whose cost is the number of scattered accesses it
// issues (30 M Gaussians: three per instance -- two table reads and the key store -- cost 0.5 ms of address
// processing; this pass moves 66 MB, coalesced).
// (band views: only the rows' entries of the band's tiles [t_lo, t_hi) exist -- lr_project_batched_kernel<true>)
__global__ void __launch_bounds__(256)
lr_rebase_kernel(uint32_t* __restrict__ state, uint32_t tiles, uint32_t batches, uint32_t t_lo, uint32_t t_hi) {
  const uint32_t* __restrict__ offsets = state + lr_offsets_off(tiles);
  uint32_t* __restrict__ row = state + lr_basetab_off(tiles) + (size_t)blockIdx.y * tiles;
  const uint32_t t = t_lo + blockIdx.x * 256u + threadIdx.x;
  if (t < t_hi) row[t] += offsets[t];
}
void lr_launch_rebase(uint32_t* state, uint32_t tiles, uint32_t batches, uint32_t t_lo, uint32_t t_hi, hipStream_t s) {
  if (!batches || t_hi <= t_lo) return;
  lr_prof_begin(LRK_REBASE, s);
  hipLaunchKernelGGL(lr_rebase_kernel, dim3((t_hi - t_lo + 255u) / 256u, batches), dim3(256), 0, s, state, tiles, batches,
                     t_lo, t_hi);
  lr_prof_end(LRK_REBASE, s);
}

// ---- A3: per-tile bucket fill ---------------------------------------------------------------------------
// key = (fp32 bits of view depth) << 32 | Gaussian index; depth > 0.2 so the bit pattern is monotone.
// (Also clears point_weight[N] and the caller's backward scratch: see the top of the kernel.)
// Gaussians with <= 4 tiles already own their slots (q3): position = offsets[tile] + slot, no atomics.  Larger
// rects take positions from the per-tile cursor; up to LR_COOP_TILES tiles a lane expands its own rect,
// beyond that the whole wave expands it (ballot over the lanes that hold one, record broadcast with
// readlane) so that a single screen-filling Gaussian does not serialise its wave.
// The forward's verdict on its buffers, reached by every workgroup of the fill from the scan's header: the caller's
// buffers hold `capacity` instances and it launched the sort levels for lists of up to `max_len_hint` keys (0 = no hint:
// the levels for `capacity`); if either is exceeded nothing may be sorted or composited.  The flag makes the later
// kernels return at once (lr_bail), and the caller's status block (optional, include/lograst.h: LOGRAST_STATUS_*) records
// this forward and the running maxima / sticky overflow bit across forwards (written by one thread of the grid).
LR_DEV bool lr_fill_verdict(uint32_t* __restrict__ state, uint32_t capacity, uint32_t max_len_hint,
                            uint32_t* __restrict__ status, int speculative, const void* keys) {
  const uint32_t total = state[LR_HDR_NUM], maxlen = state[LR_HDR_MAXLEN];
  const bool over = total > capacity || (max_len_hint != 0u && maxlen > max_len_hint);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // (written either way: a second stage-2 pass over the same tile_state with larger buffers -- the retry of a
    // speculative forward, lograst_forward_speculative -- must find the flag of the failed attempt cleared)
    state[LR_HDR_OVERFLOW] = over ? 1u : 0u;
    const uint64_t kp = (uint64_t)reinterpret_cast<uintptr_t>(keys);
    state[LR_HDR_KEYS_LO] = (uint32_t)kp; state[LR_HDR_KEYS_HI] = (uint32_t)(kp >> 32); state[LR_HDR_KEYS_CAP] = capacity;
    // a speculative attempt that overflows is repeated by the caller with exact buffers: only that pass is recorded
    if (status && !(speculative && over)) {
      status[LOGRAST_STATUS_LAST_INSTANCES] = total;
      status[LOGRAST_STATUS_LAST_OVERFLOW] = over ? 1u : 0u;
      status[LOGRAST_STATUS_LAST_MAX_LEN] = maxlen;
      status[LOGRAST_STATUS_LAST_RECT] = state[LR_HDR_RECT];
      atomicMax(&status[LOGRAST_STATUS_MAX_INSTANCES], total);
      atomicMax(&status[LOGRAST_STATUS_MAX_MAX_LEN], maxlen);
      atomicAdd(&status[LOGRAST_STATUS_FORWARDS], 1u);
      if (over) atomicOr(&status[LOGRAST_STATUS_OVERFLOW], 1u);
    }
  }
  return over;
}

// Rects of more than LR_RANKED_TILES tiles were only counted by the projection: their instances take positions from the
// per-tile cursors here.  The projection kernel's support test is repeated (same record, same code) so that the same
// tiles are filled.  Up to LR_COOP_TILES tiles a lane expands its own rect, beyond that the whole wave expands it (ballot
// over the lanes that hold one, record broadcast with readlane).  Every lane of the wave must call this (nt = 0: nothing).
// slot_of(tile, owner_id): first slot of the owner's batch in `tile` (absolute): the staged row or the table look-up.
// rmid / mrow: the rect was ranked by the batched projection (fill record bit 31): its instances go to slot + rank, the
// ranks read back from its rank row -- no cursor atomic, no support test, no record fetch.
template <typename SlotOf>
LR_DEV void lr_fill_big_rect(const float4* __restrict__ geom, int i, int x0, int y0, int w, int h, int nt, uint64_t key,
                             int gx, bool tile_cull, uint32_t* __restrict__ cursor, uint64_t* __restrict__ keys, int lane,
                             int mid_coop, bool rmid, int mrow, const uint16_t* __restrict__ midrank, uint32_t batch,
                             SlotOf&& slot_of LR_ABLATE_PARAM) {
  const int y1 = y0 + h, x1 = x0 + w;
  LrSupport sup = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1};  // mode 1: every tile of the rect
  if (nt > LR_RANKED_TILES && tile_cull && !rmid) {
    const float4 g0 = geom[LR_REC_QUADS * (size_t)i + 0], g1 = geom[LR_REC_QUADS * (size_t)i + 1];
    sup = lr_support_prepare(g0.x, g0.y, g0.z, g0.w, g1.x, g1.y);
  }
  const int klo = (int)(uint32_t)key, khi = (int)(uint32_t)(key >> 32);
  if (__builtin_amdgcn_ballot_w64(rmid) != 0) {   // (wave-uniform) ranked rects: slot of the batch's run + rank
    const uint32_t midcap = lr_mid_cap(batch);
    lr_mid_rects<false>(rmid, x0, y0, w, nt, sup, mrow, klo, khi, [&](int t, int ty, int tx, bool, int mr, int lo, int hi) {
      const uint32_t b = (uint32_t)lo / batch;             // (the key's low word is the Gaussian's index)
      const uint32_t r = midrank[((size_t)b * midcap + (uint32_t)mr) * LR_MID_ROW + t];
      if (r != 0xffffu) keys[slot_of(ty * gx + tx, (uint32_t)lo) + r] = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
    });
  }
  const bool mid = !rmid && nt > LR_RANKED_TILES && nt <= LR_COOP_TILES && !LR_ABLATED(4);
  if (mid_coop) {   // the wave expands its 5..16-tile rects together, four per pass (lr_mid_rects)
    if (__builtin_amdgcn_ballot_w64(mid) != 0) {
      lr_mid_rects<true>(mid, x0, y0, w, nt, sup, 0, klo, khi, [&](int, int ty, int tx, bool keep, int, int lo, int hi) {
        if (keep) {
          const uint32_t pos = atomicAdd(&cursor[(ty * gx + tx) * LR_CTR_STRIDE], 1u);
          keys[pos] = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
        }
      });
    }
  } else if (mid) {
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++)
        if (lr_support_tile(sup, x, y)) {
          uint32_t pos = atomicAdd(&cursor[(y * gx + x) * LR_CTR_STRIDE], 1u);
          keys[pos] = key;
        }
  }
  uint64_t bigm = __ballot(nt > LR_COOP_TILES && !LR_ABLATED(8));
  while (bigm) {
    int src = __builtin_ctzll(bigm);
    bigm &= bigm - 1;
    int bx0 = lr_readlane_i(x0, src), by0 = lr_readlane_i(y0, src);
    int bw = lr_readlane_i(w, src), bn = lr_readlane_i(nt, src);
    uint32_t klo = (uint32_t)lr_readlane_i((int)(uint32_t)key, src);
    uint32_t khi = (uint32_t)lr_readlane_i((int)(uint32_t)(key >> 32), src);
    uint64_t bkey = ((uint64_t)khi << 32) | klo;
    LrSupport bs;
    bs.mx = lr_readlane_f(sup.mx, src); bs.my = lr_readlane_f(sup.my, src);
    bs.A = lr_readlane_f(sup.A, src); bs.B = lr_readlane_f(sup.B, src); bs.C = lr_readlane_f(sup.C, src);
    bs.tau = lr_readlane_f(sup.tau, src); bs.ex = lr_readlane_f(sup.ex, src); bs.ey = lr_readlane_f(sup.ey, src);
    bs.iA = lr_readlane_f(sup.iA, src); bs.iC = lr_readlane_f(sup.iC, src);
    bs.mode = lr_readlane_i(sup.mode, src);
    for (int t = lane; t < bn; t += 64) {
      int ty = t / bw, tx = t - ty * bw;
      if (lr_support_tile(bs, bx0 + tx, by0 + ty)) {
        uint32_t pos = atomicAdd(&cursor[((by0 + ty) * gx + (bx0 + tx)) * LR_CTR_STRIDE], 1u);
        keys[pos] = bkey;
      }
    }
  }
}

template <int K>   // Gaussians per thread: their fill records are requested together (see lr_launch_fill)
__global__ void __launch_bounds__(256)
lr_fill_kernel(int N, int gx, const float4* __restrict__ geom, uint32_t* __restrict__ state, uint32_t tiles,
               uint64_t* __restrict__ keys, uint32_t capacity, uint32_t max_len_hint, uint32_t* __restrict__ status,
               float* __restrict__ zero_n, float* __restrict__ zero_block, int zero_block_floats, int xcd_order, int stream_nt,
               int rebased, int speculative, int mid_coop LR_ABLATE_PARAM) {
  // Per-Gaussian buffers that later kernels accumulate into with atomics (point_weight; the backward scratch)
  // are cleared here, in a kernel that already has one thread per Gaussian, instead of by separate memsets.
  // XCD-contiguous block order (speed only): blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8,
  // observed, MI355X_MICROARCH.md), each with its own non-coherent L2.  In dispatch order the 8-byte key writes of one
  // (batch, tile) segment -- adjacent in memory, written by Gaussians of the same batch -- would be spread over all
  // eight L2s and leave each of them as a partial line (a read-modify-write at HBM: 30 M Gaussians, 44 M keys took
  // 1.1 ms = 128 B of traffic per key).  With XCD x walking Gaussians [x N/8, (x+1) N/8) the segments of consecutive
  // batches complete their lines inside one L2 before they are evicted.
  // K > 1: a workgroup owns K blocks of 256 Gaussians that lie a K-th of its XCD's share apart -- K streams per XCD, each
  // walking consecutive Gaussians like the single one (K ADJACENT blocks per workgroup lost on the 30 M view: 433 / 462 /
  // 493 us for K = 1 / 2 / 4).
  const uint32_t per_xcd_sub = gridDim.x >> 3;                   // grid is a multiple of 8: blocks per XCD and stream
  const uint32_t per_xcd = per_xcd_sub * K;
  uint32_t vblock_k[K];
#pragma unroll
  for (int u = 0; u < K; u++)
    vblock_k[u] = xcd_order ? (blockIdx.x & 7u) * per_xcd + u * per_xcd_sub + (blockIdx.x >> 3) : blockIdx.x + u * gridDim.x;
  {
    // (streaming stores: non-temporal, so that they do not push the partially written key lines of this kernel out
    // of the XCD's L2 before their neighbours arrive)
#pragma unroll
    for (int u = 0; u < K; u++) {
      const int zi = (int)(vblock_k[u] * 256u + threadIdx.x);
      if (zi < N && !LR_ABLATED(1)) {
        if (stream_nt) {
          if (zero_n) __builtin_nontemporal_store(0.f, &zero_n[zi]);
          for (int k = 0; k < zero_block_floats; k++) __builtin_nontemporal_store(0.f, &zero_block[(size_t)k * N + zi]);
        } else {
          if (zero_n) zero_n[zi] = 0.f;
          for (int k = 0; k < zero_block_floats; k++) zero_block[(size_t)k * N + zi] = 0.f;
        }
      }
    }
  }
  const bool tile_cull = state[LR_HDR_CULL] != 0u;
  const bool over = lr_fill_verdict(state, capacity, max_len_hint, status, speculative, keys);
  if (over) return;
  const uint32_t* __restrict__ offsets = state + lr_offsets_off(tiles);
  uint32_t* cursor = state + lr_cursor_off(tiles);
  // batched projection: a ranked instance's slot is relative to its batch's reservation in the tile
  const uint32_t batch = state[LR_HDR_BATCH];
  const int lane = threadIdx.x & 63;
  const uint4* __restrict__ fillrec = reinterpret_cast<const uint4*>(geom + LR_REC_QUADS * (size_t)N);
  // The K fill records of this thread, requested together: a thread's work is a chain of dependent accesses (header,
  // fill record, slot table, key store) that a workgroup waits through once, whatever K (LOGRAST_FILL_PER_THREAD).
  bool vis_k[K];
  uint4 fr_k[K];
  int id_k[K];
  // band views (lr_project_band_kernel): projection workgroup w left the fill records of its survcount[w] survivors in
  // slots [w * span, ...), their Gaussian indices in the same slots of the array behind the fill records
  const bool sparse = batch && state[LR_HDR_SPARSE] != 0u;
  const uint32_t* __restrict__ survivor = reinterpret_cast<const uint32_t*>(fillrec + N);
  const uint16_t* __restrict__ midrank =
      reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(geom) + lr_midrank_off_bytes((size_t)N));
  const uint32_t span = sparse ? state[LR_HDR_SPAN] : 1u;
  const uint32_t* __restrict__ survcount =
      state + lr_survcount_off(tiles, batch ? ((uint32_t)N + batch - 1u) / batch : 0u);
#pragma unroll
  for (int u = 0; u < K; u++) {
    const uint32_t e = vblock_k[u] * 256u + threadIdx.x;                 // slot (= Gaussian unless sparse)
    bool vis = e < (uint32_t)N;
    if (vis && sparse) { const uint32_t w = e / span; vis = e - w * span < survcount[w]; }
    fr_k[u] = uint4{0u, 0xffffffffu, 0u, 0u};
    id_k[u] = (int)e;
    if (vis && batch) {
      if (stream_nt) {
        typedef uint32_t lr_u4v __attribute__((ext_vector_type(4)));
        const lr_u4v t4 = __builtin_nontemporal_load(reinterpret_cast<const lr_u4v*>(fillrec + e));
        fr_k[u] = uint4{t4.x, t4.y, t4.z, t4.w};
      } else {
        fr_k[u] = fillrec[e];
      }
      if (sparse) id_k[u] = (int)survivor[e];
    }
    vis_k[u] = vis;
  }
#pragma unroll
  for (int u = 0; u < K; u++) {
  const int i = id_k[u];
  const bool vis = vis_k[u];
  const uint32_t* __restrict__ bbase =
      batch ? state + lr_basetab_off(tiles) + (size_t)((uint32_t)i / batch) * tiles : nullptr;
  uint32_t dbits = 0;
  int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  bool rmid = false;                                           // a 5..16-tile rect the projection ranked: fr.w = its rank row
  int mrow = 0;
  uint32_t slot[LR_RANKED_TILES] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  if (vis && batch) {
    // batched projection: the 16-byte fill record (see lr_project_batched_kernel)
    const uint4 fr = fr_k[u];
    dbits = fr.x;
    if (fr.y != 0xffffffffu) {
      x0 = (int)(fr.y & 0x1fffu); y0 = (int)((fr.y >> 13) & 0x1fffu);
      if (fr.y & (1u << 30)) {
        x1 = (int)(fr.z & 0xffffu); y1 = (int)(fr.z >> 16);
        rmid = (fr.y >> 31) != 0u; mrow = (int)fr.w;
      } else {
        x1 = x0 + (int)((fr.y >> 26) & 3u) + 1; y1 = y0 + (int)((fr.y >> 28) & 3u) + 1;
        const uint32_t h0 = fr.z & 0xffffu, h1 = fr.z >> 16, h2 = fr.w & 0xffffu, h3 = fr.w >> 16;
        slot[0] = h0 == 0xffffu ? 0xffffffffu : h0; slot[1] = h1 == 0xffffu ? 0xffffffffu : h1;
        slot[2] = h2 == 0xffffu ? 0xffffffffu : h2; slot[3] = h3 == 0xffffu ? 0xffffffffu : h3;
      }
    }
  } else if (vis) {
    const float4 g2 = geom[LR_REC_QUADS * (size_t)i + 2];
    dbits = __float_as_uint(g2.y);
    const uint32_t r0 = __float_as_uint(g2.z), r1 = __float_as_uint(g2.w);
    x0 = (int)(r0 & 0xffff); y0 = (int)(r0 >> 16); x1 = (int)(r1 & 0xffff); y1 = (int)(r1 >> 16);
    if ((x1 - x0) * (y1 - y0) <= LR_RANKED_TILES && (x1 - x0) * (y1 - y0) > 0) {
      const float4 g3 = geom[LR_REC_QUADS * (size_t)i + 3];
      slot[0] = __float_as_uint(g3.x); slot[1] = __float_as_uint(g3.y);
      slot[2] = __float_as_uint(g3.z); slot[3] = __float_as_uint(g3.w);
    }
  }
  int w = x1 - x0, h = y1 - y0;
  int nt = vis ? w * h : 0;
  uint64_t key = ((uint64_t)dbits << 32) | (uint32_t)i;
  if (nt > 0 && nt <= LR_RANKED_TILES) {
#pragma unroll
    for (int k = 0; k < LR_RANKED_TILES; k++) {
      if (k < nt && slot[k] != 0xffffffffu) {  // 0xffffffff: dropped by the support cull in the projection kernel
        const int ty = (w == 1) ? k : ((w == 2 && nt == 4) ? (k >> 1) : 0), tx = k - ty * w;  // as in lr_project_one
        const int t = (y0 + ty) * gx + (x0 + tx);
        // batched: slot relative to the batch's run in the tile; lr_rebase_kernel (large inputs) made the table absolute
        const uint32_t pos = (batch ? (rebased ? bbase[t] : offsets[t] + bbase[t]) : offsets[t]) + slot[k];
        if (!LR_ABLATED(2) || pos == 0xffffffffu) keys[pos] = key;
      }
    }
  }
  lr_fill_big_rect(geom, i, x0, y0, w, h, nt, key, gx, tile_cull, cursor, keys, lane, mid_coop, rmid, mrow, midrank, batch,
                   [&](int t, uint32_t owner) -> uint32_t {
                     const uint32_t* __restrict__ bb = state + lr_basetab_off(tiles) + (size_t)(owner / batch) * tiles;
                     return rebased ? bb[t] : offsets[t] + bb[t];
                   } LR_ABLATE_PASS(ablate));
  }
}

// ---- A3, batched full views: the batch's slot table staged in LDS ------------------------------------------------------
// What the fill costs is the number of scattered accesses it sends to the L2s: per tile instance one 4-byte look-up in the
// batch's row of the slot table and one 8-byte key store -- 88 M requests for the 30 M-Gaussian view's 44 M instances,
// against ~270 G requests/s that the L2 channels accept chip-wide (0.33 of the kernel's 0.39 ms).  Here a workgroup owns
// LR_FILL_STAGED_ROWS CONSECUTIVE Gaussians -- one batch: batches are multiples of 1024 -- and first copies that batch's
// table row (tiles x 4 B: 32 KB at 1080p; `+ offsets[t]` on inputs too small for lr_rebase_kernel) into LDS with coalesced
// 16-byte loads: 256 line requests per workgroup (L2 hits: the ~29 workgroups of a batch run on one XCD within
// microseconds of each other) instead of ~1500 scattered look-ups.  Rects of more than 4 tiles as in lr_fill_kernel.
template <int K>   // K x 1024 consecutive Gaussians per workgroup (K Gaussians per thread, their fill records requested together)
__global__ void __launch_bounds__(LR_FILL_STAGED_ROWS, 8)   // 64 VGPRs: 2048 threads per CU
lr_fill_staged_kernel(int N, int gx, const float4* __restrict__ geom, uint32_t* __restrict__ state, uint32_t tiles,
                      uint64_t* __restrict__ keys, uint32_t capacity, uint32_t max_len_hint, uint32_t* __restrict__ status,
                      float* __restrict__ zero_n, float* __restrict__ zero_block, int zero_block_floats, int xcd_order,
                      int rebased, int speculative, int mid_coop LR_ABLATE_PARAM) {
  extern __shared__ uint32_t lr_slot_row[];                       // [tiles]: absolute first slot of this batch's run in every tile
  const uint32_t per_xcd = gridDim.x >> 3;                        // grid is a multiple of 8 (XCD-contiguous order: lr_fill_kernel)
  const uint32_t vblock = xcd_order ? (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
  const uint32_t first = vblock * (LR_FILL_STAGED_ROWS * K);      // the workgroup's first Gaussian (< N unless the grid's padding)
  const bool tile_cull = state[LR_HDR_CULL] != 0u;
  const bool over = lr_fill_verdict(state, capacity, max_len_hint, status, speculative, keys);
  const uint32_t batch = state[LR_HDR_BATCH];
  const uint4* __restrict__ fillrec = reinterpret_cast<const uint4*>(geom + LR_REC_QUADS * (size_t)N);
  uint4 fr_k[K];
#pragma unroll
  for (int u = 0; u < K; u++) {
    const uint32_t e = first + u * LR_FILL_STAGED_ROWS + threadIdx.x;
    fr_k[u] = uint4{0u, 0xffffffffu, 0u, 0u};
    if (e < (uint32_t)N) {
      if (!LR_ABLATED(1)) {                                       // (small inputs: the zero-fills live here, see lr_fill_kernel)
        if (zero_n) zero_n[e] = 0.f;
        for (int k = 0; k < zero_block_floats; k++) zero_block[(size_t)k * N + e] = 0.f;
      }
      if (!over) {
        typedef uint32_t lr_u4v __attribute__((ext_vector_type(4)));
        const lr_u4v t4 = __builtin_nontemporal_load(reinterpret_cast<const lr_u4v*>(fillrec + e));
        fr_k[u] = uint4{t4.x, t4.y, t4.z, t4.w};
      }
    }
  }
  if (over) return;
  {
    const uint32_t b = (first < (uint32_t)N ? first : 0u) / batch;   // (batches are multiples of K x 1024 Gaussians: lr_launch_fill)
    const uint32_t* __restrict__ row = state + lr_basetab_off(tiles) + (size_t)b * tiles;
    const uint32_t* __restrict__ off = state + lr_offsets_off(tiles);
    if (((tiles | lr_basetab_off(tiles) | lr_offsets_off(tiles)) & 3u) == 0u) {   // rows start on 16-byte boundaries (1080p: yes)
      const uint4* __restrict__ row4 = reinterpret_cast<const uint4*>(row);
      const uint4* __restrict__ off4 = reinterpret_cast<const uint4*>(off);
      uint4* dst4 = reinterpret_cast<uint4*>(lr_slot_row);
      for (uint32_t t4 = threadIdx.x; t4 < (tiles >> 2); t4 += LR_FILL_STAGED_ROWS) {
        uint4 r = row4[t4];
        if (!rebased) { const uint4 o = off4[t4]; r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }
        dst4[t4] = r;
      }
    } else {
      for (uint32_t t = threadIdx.x; t < tiles; t += LR_FILL_STAGED_ROWS) lr_slot_row[t] = row[t] + (rebased ? 0u : off[t]);
    }
  }
  uint32_t* cursor = state + lr_cursor_off(tiles);
  const int lane = threadIdx.x & 63;
  int x0_k[K], y0_k[K], w_k[K], h_k[K], nt_k[K];
  uint32_t hA_k[K], hB_k[K];                                     // the four 16-bit ranks of a rect of <= 4 tiles (hB: the rank row of a ranked 5..16-tile rect)
  bool rmid_k[K];
  const uint16_t* __restrict__ midrank =
      reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(geom) + lr_midrank_off_bytes((size_t)N));
#pragma unroll
  for (int u = 0; u < K; u++) {
    const uint32_t e = first + u * LR_FILL_STAGED_ROWS + threadIdx.x;
    const uint4 fr = fr_k[u];
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    hA_k[u] = 0xffffffffu; hB_k[u] = 0xffffffffu;
    rmid_k[u] = false;
    if (fr.y != 0xffffffffu) {
      x0 = (int)(fr.y & 0x1fffu); y0 = (int)((fr.y >> 13) & 0x1fffu);
      if (fr.y & (1u << 30)) {
        x1 = (int)(fr.z & 0xffffu); y1 = (int)(fr.z >> 16);
        rmid_k[u] = (fr.y >> 31) != 0u; hB_k[u] = fr.w;
      } else {
        x1 = x0 + (int)((fr.y >> 26) & 3u) + 1; y1 = y0 + (int)((fr.y >> 28) & 3u) + 1;
        hA_k[u] = fr.z; hB_k[u] = fr.w;
      }
    }
    x0_k[u] = x0; y0_k[u] = y0; w_k[u] = x1 - x0; h_k[u] = y1 - y0;
    nt_k[u] = e < (uint32_t)N ? w_k[u] * h_k[u] : 0;
  }
  __syncthreads();                                                // the staged row is complete
#pragma unroll
  for (int u = 0; u < K; u++) {
    const int i = (int)(first + u * LR_FILL_STAGED_ROWS + threadIdx.x);
    const int x0 = x0_k[u], y0 = y0_k[u], w = w_k[u], nt = nt_k[u];
    const uint64_t key = ((uint64_t)fr_k[u].x << 32) | (uint32_t)i;
    if (nt > 0 && nt <= LR_RANKED_TILES) {
      const uint32_t h0 = hA_k[u] & 0xffffu, h1 = hA_k[u] >> 16, h2 = hB_k[u] & 0xffffu, h3 = hB_k[u] >> 16;
      // tile k of a rect of <= 4 tiles: one row (w >= nt), one column (w == 1) or 2x2 -- as in lr_project_batched_kernel
      const bool col = w == 1, sq = (w == 2) && (nt == 4);
      const int t0 = y0 * gx + x0;
      const int d1 = col ? gx : 1, d2 = col ? 2 * gx : (sq ? gx : 2), d3 = col ? 3 * gx : (sq ? gx + 1 : 3);
      // all look-ups (LDS) before the first store
      const uint32_t p0 = h0 != 0xffffu ? lr_slot_row[t0] + h0 : 0xffffffffu;
      const uint32_t p1 = (nt > 1 && h1 != 0xffffu) ? lr_slot_row[t0 + d1] + h1 : 0xffffffffu;
      const uint32_t p2 = (nt > 2 && h2 != 0xffffu) ? lr_slot_row[t0 + d2] + h2 : 0xffffffffu;
      const uint32_t p3 = (nt > 3 && h3 != 0xffffu) ? lr_slot_row[t0 + d3] + h3 : 0xffffffffu;
      if (!LR_ABLATED(2)) {
        if (p0 != 0xffffffffu) keys[p0] = key;
        if (p1 != 0xffffffffu) keys[p1] = key;
        if (p2 != 0xffffffffu) keys[p2] = key;
        if (p3 != 0xffffffffu) keys[p3] = key;
      }
    }
    // (Rects of more than four tiles, measured and removed: counting the workgroup's instances per tile in LDS first, ONE
    // cursor atomic per touched tile, a second walk placing the keys -- the C3 view's fill 426 -> 545 us: 2048 consecutive
    // rows of a level-of-detail selection do not share enough tiles to pay for two walks of support tests.)
    lr_fill_big_rect(geom, i, x0, y0, w, h_k[u], nt, key, gx, tile_cull, cursor, keys, lane, mid_coop, rmid_k[u], (int)hB_k[u],
                     midrank, batch, [&](int t, uint32_t) -> uint32_t { return lr_slot_row[t]; } LR_ABLATE_PASS(ablate));
  }
}

void lr_launch_fill(int N, int gx, const void* geom, uint32_t* state, uint32_t tiles, uint64_t* keys,
                    uint32_t capacity, uint32_t max_len_hint, uint32_t* status, float* zero_n, float* zero_block,
                    int zero_block_floats, int rebased, int speculative, int band, int staged_k, hipStream_t s) {
  if (N <= 0) return;
  lr_prof_begin(LRK_FILL, s);
  LR_KNOB(xcd_order, "LOGRAST_FILL_XCD_ORDER", 1);
  LR_KNOB(fill_nt, "LOGRAST_FILL_NT", 1);
  LR_KNOB(mid_coop, "LOGRAST_MID_COOP", 16);
#ifdef LR_EXPERIMENTS
  static const int ablate = lr_env_int("LOGRAST_FILL_ABLATE", 0);   // timing experiments (tools/): 1 no zero-fill, 2 no key stores, 4 no 5-16-tile rects, 8 no larger rects
#endif
  if (staged_k > 0) {   // (decided by the caller -- api.hip: lr_fill_staged_k -- because stage 1 has to know it too: no lr_rebase_kernel then)
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lr_fill_staged_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, sizeof(uint32_t) * LR_FILL_STAGED_MAX_TILES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lr_fill_staged_kernel<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, sizeof(uint32_t) * LR_FILL_STAGED_MAX_TILES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lr_fill_staged_kernel<3>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, sizeof(uint32_t) * LR_FILL_STAGED_MAX_TILES);
      attr_set = true;
    }
    const int K = staged_k;
    const int rows = LR_FILL_STAGED_ROWS * K;
    const int blocks = ((N + rows - 1) / rows + 7) & ~7;
#define LR_FILL_ST(KK) hipLaunchKernelGGL(lr_fill_staged_kernel<KK>, dim3(blocks), dim3(LR_FILL_STAGED_ROWS),                 \
                       sizeof(uint32_t) * ((tiles + 3u) & ~3u), s, N, gx, reinterpret_cast<const float4*>(geom), state, tiles,  \
                       keys, capacity, max_len_hint, status, zero_n, zero_block, zero_block_floats, xcd_order, rebased,         \
                       speculative, mid_coop LR_ABLATE_PASS(ablate))
    if (K >= 3) LR_FILL_ST(3); else if (K == 2) LR_FILL_ST(2); else LR_FILL_ST(1);   // (four per thread: 40 bytes of scratch at the kernel's 64 VGPRs)
#undef LR_FILL_ST
    lr_prof_end(LRK_FILL, s);
    return;
  }
  LR_KNOB(per_thread_knob, "LOGRAST_FILL_PER_THREAD", 1);
  int per_thread = per_thread_knob;
#define LR_FILL(K) do { const int blocks = (((N + 255) / 256 + K - 1) / K + 7) & ~7;                                     \
    hipLaunchKernelGGL(lr_fill_kernel<K>, dim3(blocks), dim3(256), 0, s, N, gx, reinterpret_cast<const float4*>(geom),  \
                       state, tiles, keys, capacity, max_len_hint, status, zero_n, zero_block, zero_block_floats,       \
                       xcd_order, fill_nt, rebased, speculative, mid_coop LR_ABLATE_PASS(ablate)); } while (0)
  // measured, K = 1 / 2 / 4: the 30 M view 388 / 413 / 424 us; a band view (100 M, a fifth of the slots used: most
  // workgroups only pass through the chain once) 618 / 551 / 510 us
  if (band) per_thread = 4;
  if (per_thread >= 4) LR_FILL(4); else if (per_thread >= 2) LR_FILL(2); else LR_FILL(1);
#undef LR_FILL
  lr_prof_end(LRK_FILL, s);
}

// Clears the header + per-tile counters of a tile_state before the projection (a kernel rather than hipMemsetAsync: the
// launch sequence of a view is captured into HIP graphs, several of which replay concurrently on different streams, and
// memset nodes of concurrently replayed graphs were observed to leave the counters of one of them uncleared).
__global__ void __launch_bounds__(256)
lr_zero_words_kernel(uint4* __restrict__ p, uint32_t n4) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n4) p[i] = uint4{0u, 0u, 0u, 0u};
}
// n floats at any 4-byte-aligned address, streamed: at 30 M Gaussians the forward's zero-fills (point_weight and the
// backward's accumulators: 480 MB) take 72 us on their own and 131 us when the fill kernel issues them between its
// scattered accesses (that fusion pays below ~4 M Gaussians, where a launch costs more than the stores).
__global__ void __launch_bounds__(256)
lr_zero_floats_kernel(float* __restrict__ p, size_t n) {
  const size_t head = min(n, (size_t)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(p) & 15u)) & 15u) >> 2));
  const size_t n4 = (n - head) >> 2;
  typedef uint32_t lr_u4v __attribute__((ext_vector_type(4)));
  lr_u4v* q = reinterpret_cast<lr_u4v*>(p + head);
  const lr_u4v z = {0u, 0u, 0u, 0u};
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256u)
    __builtin_nontemporal_store(z, q + i);
  if (blockIdx.x == 0 && threadIdx.x < 8u) {                 // the unaligned head and the tail
    const size_t t = threadIdx.x < 4u ? threadIdx.x : head + 4u * n4 + (threadIdx.x - 4u);
    if (threadIdx.x < 4u ? t < head : t < n) p[t] = 0.f;
  }
}
void lr_launch_zero_floats(float* p, size_t n, hipStream_t s) {
  if (!p || !n) return;
  const size_t blocks = min((n / 4 + 255) / 256 + 1, (size_t)16384);
  hipLaunchKernelGGL(lr_zero_floats_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, p, n);
}
void lr_launch_zero_words(uint32_t* p, size_t words, hipStream_t s) {
  const uint32_t n4 = (uint32_t)((words + 3) / 4);   // callers pad to 16 bytes (LR_HDR_WORDS and the counter arrays are multiples of 4 words)
  if (!n4) return;
  hipLaunchKernelGGL(lr_zero_words_kernel, dim3((n4 + 255u) / 256u), dim3(256), 0, s, reinterpret_cast<uint4*>(p), n4);
}

// ---- measured roof: a streaming device-to-device copy (bench.py's HBM denominator) --------------------------------
// 16 bytes per lane per access: the float4 copy /opt/skills/guides/MI355X_MICROARCH.md quotes at 6.29 TB/s (read + write
// counted).  Not on the rasterizer's path; exported as lograst_stream_copy so that bench.py divides by a rate this
// library's own code reaches on the same box.  Forms (bits 20+ of the `blocks` argument; bench.py takes the best of all):
//   0  grid-stride, four independent non-temporal loads in flight per lane, non-temporal stores
//   1  one access per lane, no loop (grid = n16 / 256), plain loads and stores
//   2  grid-stride, eight non-temporal loads in flight per lane, non-temporal stores
//   3  grid-stride x 4, plain loads, non-temporal stores
//   4  one access per lane, no loop, non-temporal both ways
typedef uint32_t lr_u4v __attribute__((ext_vector_type(4)));
template <int U, bool NT_LOAD, bool NT_STORE>
__global__ void __launch_bounds__(256)
lr_stream_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  const lr_u4v* __restrict__ a = reinterpret_cast<const lr_u4v*>(src);
  lr_u4v* __restrict__ b = reinterpret_cast<lr_u4v*>(dst);
  const size_t stride = (size_t)gridDim.x * 256u;
  size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  for (; i + (U - 1) * stride < n16; i += U * stride) {   // U independent 16-byte loads in flight per lane
    lr_u4v v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = NT_LOAD ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (NT_STORE) __builtin_nontemporal_store(v[u], b + i + u * stride); else b[i + u * stride] = v[u];
    }
  }
  for (; i < n16; i += stride) {
    const lr_u4v t = NT_LOAD ? __builtin_nontemporal_load(a + i) : a[i];
    if (NT_STORE) __builtin_nontemporal_store(t, b + i); else b[i] = t;
  }
}
template <bool NT>
__global__ void __launch_bounds__(256)
lr_stream_copy_flat_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  const lr_u4v* __restrict__ a = reinterpret_cast<const lr_u4v*>(src);
  lr_u4v* __restrict__ b = reinterpret_cast<lr_u4v*>(dst);
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (i < n16) {
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i); else b[i] = a[i];
  }
}
void lr_launch_stream_copy(const void* src, void* dst, size_t bytes, int blocks, hipStream_t s) {
  const size_t n16 = bytes / 16;
  if (!n16) return;
  const int mode = blocks > 0 ? blocks >> 20 : 0;
  blocks = blocks > 0 ? (blocks & 0xfffff) : 0;
  if (blocks <= 0) blocks = 256 * 16;   // 16 workgroups of 256 per CU
  const uint4* a = reinterpret_cast<const uint4*>(src);
  uint4* b = reinterpret_cast<uint4*>(dst);
  const uint32_t flat = (uint32_t)((n16 + 255) / 256);
  switch (mode) {
    case 1: hipLaunchKernelGGL(lr_stream_copy_flat_kernel<false>, dim3(flat), dim3(256), 0, s, a, b, n16); break;
    case 4: hipLaunchKernelGGL(lr_stream_copy_flat_kernel<true>, dim3(flat), dim3(256), 0, s, a, b, n16); break;
    case 2: hipLaunchKernelGGL((lr_stream_copy_kernel<8, true, true>), dim3((uint32_t)blocks), dim3(256), 0, s, a, b, n16); break;
    case 3: hipLaunchKernelGGL((lr_stream_copy_kernel<4, false, true>), dim3((uint32_t)blocks), dim3(256), 0, s, a, b, n16); break;
    default: hipLaunchKernelGGL((lr_stream_copy_kernel<4, true, true>), dim3((uint32_t)blocks), dim3(256), 0, s, a, b, n16); break;
  }
}

// ---- image split into bands of tile rows: which rows does each Gaussian reach? -------------------
// The rect's tile-row range [y0, y1) of every Gaussian over the WHOLE image (y0 | y1 << 16; 0 = culled / empty rect),
// from the very code the projection runs (lr_project_one with counters that count nothing), so that "rows [b, e)
// intersect [y0, y1)" is exactly "the projection clipped to the band [b, e) keeps this Gaussian".  A rank that owns a
// band selects its Gaussians from this array and projects, bins, composites and differentiates only those: 44 bytes
// read + 4 written per Gaussian here instead of the full projection (and the full chain rule) over all of them.
struct LrNoCounters {
  LR_DEV uint32_t rank(int) const { return 0u; }
  LR_DEV void count_big(int) const {}
};
__global__ void __launch_bounds__(256)
lr_tile_rows_kernel(LrView v, int N, const float* __restrict__ means, const float* __restrict__ scales,
                    const float* __restrict__ rots, uint32_t* __restrict__ rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  LrInputs in;
  in.p[0] = means[3 * i]; in.p[1] = means[3 * i + 1]; in.p[2] = means[3 * i + 2];
  if (v.cov3d) {
    const float* __restrict__ c6 = v.cov3d + 6 * (size_t)i;
    in.s[0] = c6[0]; in.s[1] = c6[1]; in.s[2] = c6[2];
    in.q = float4{c6[3], c6[4], c6[5], 0.f};
  } else {
    in.s[0] = scales[3 * i]; in.s[1] = scales[3 * i + 1]; in.s[2] = scales[3 * i + 2];
    in.q = reinterpret_cast<const float4*>(rots)[i];
  }
  in.op = 1.f; in.c[0] = 0.f; in.c[1] = 0.f; in.c[2] = 0.f;
  float4 g0, g1, g2, g3;
  int rad = 0;
  uint32_t rect = 0;
  bool huge = false;
  lr_project_one<true>(v, in, 0, LrNoCounters{}, g0, g1, g2, g3, rad, rect, huge, 0);
  rows[i] = rad > 0 ? ((__float_as_uint(g2.z) >> 16) | (__float_as_uint(g2.w) & 0xffff0000u)) : 0u;
}
void lr_launch_tile_rows(const LrView& v, int N, const float* means, const float* scales, const float* rots,
                         uint32_t* rows, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(lr_tile_rows_kernel, dim3((N + 255) / 256), dim3(256), 0, s, v, N, means, scales, rots, rows);
}

// knn.hip -- "next" row N1: replacement for simple_knn._C.distCUDA2, the only other native op
// apps/train.py needs before the first step (/root/reference/LoG/utils/file.py:88-91,
// LoG/model/base_gaussian.py:39-42): for every point, the MEAN SQUARED DISTANCE TO ITS 3 NEAREST OTHER
// POINTS (initial Gaussian scales).  Exact 3-NN, fp32.
//
// Pipeline (all on the caller's stream, no host sync, no allocation):
//   1. bounding box of the cloud (block reduce + ordered-int atomics)
//   2. 30-bit Morton code per point (10 bits/axis inside the box)
//   3. rocPRIM radix sort of (code, index) pairs -- the one library primitive used (a plain key/value sort)
//   4. gather the points into Morton order; min/max box of every run of 1024 consecutive points
//   5. one thread per (sorted) point: seed the 3 best squared distances from its +-3 Morton neighbours, then
//      visit every 1024-point box whose distance to the point is below the current 3rd best.  A workgroup
//      holds 256 Morton-consecutive points, so its lanes want the same few boxes: a box is staged once in LDS
//      (12 KB) when ANY lane of the workgroup needs it (__syncthreads_or), and skipped by the whole workgroup
//      otherwise.  Pruning is conservative, so the result is the exact 3-NN.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

#define LR_KNN_BOX 1024

LR_DEV uint32_t lr_ord(float f) {  // monotone float -> uint map (for atomicMin/Max)
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float lr_unord(uint32_t u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  union { uint32_t u; float f; } c;
  c.u = u;
  return c.f;
}

// bbox[0..2] = ordered min, bbox[3..5] = ordered max (initialised to 0xffffffff / 0 by the launcher)
__global__ void __launch_bounds__(256)
lr_knn_bbox_kernel(int P, const float* __restrict__ pts, uint32_t* __restrict__ bbox) {
  __shared__ uint32_t smin[3], smax[3];
  if (threadIdx.x < 3) { smin[threadIdx.x] = 0xffffffffu; smax[threadIdx.x] = 0u; }
  __syncthreads();
  uint32_t mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      uint32_t o = lr_ord(pts[3 * (size_t)i + a]);
      mn[a] = min(mn[a], o);
      mx[a] = max(mx[a], o);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) { atomicMin(&smin[a], mn[a]); atomicMax(&smax[a], mx[a]); }
  __syncthreads();
  if (threadIdx.x < 3) { atomicMin(&bbox[threadIdx.x], smin[threadIdx.x]); atomicMax(&bbox[3 + threadIdx.x], smax[threadIdx.x]); }
}

LR_DEV uint32_t lr_spread3(uint32_t x) {  // 10 bits -> every third bit
  x = (x | (x << 16)) & 0x030000FFu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}

__global__ void __launch_bounds__(256)
lr_knn_morton_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ bbox,
                     uint32_t* __restrict__ codes, uint32_t* __restrict__ idx) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  uint32_t c = 0;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float lo = lr_unord(bbox[a]), hi = lr_unord(bbox[3 + a]);
    const float ext = fmaxf(hi - lo, 1e-30f);
    float t = (pts[3 * (size_t)i + a] - lo) / ext * 1023.f;
    uint32_t q = (uint32_t)fminf(fmaxf(t, 0.f), 1023.f);
    c |= lr_spread3(q) << a;
  }
  codes[i] = c;
  idx[i] = (uint32_t)i;
}

// sorted points (x,y,z,original index as bits) + per-box bounds (6 floats per box)
__global__ void __launch_bounds__(256)
lr_knn_gather_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ sidx,
                     float4* __restrict__ spts, float* __restrict__ boxes) {
  __shared__ uint32_t smin[3], smax[3];
  if (threadIdx.x < 3) { smin[threadIdx.x] = 0xffffffffu; smax[threadIdx.x] = 0u; }
  __syncthreads();
  const int base = blockIdx.x * LR_KNN_BOX;
  uint32_t mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
  for (int k = threadIdx.x; k < LR_KNN_BOX; k += 256) {
    const int i = base + k;
    if (i < P) {
      const uint32_t src = sidx[i];
      const float x = pts[3 * (size_t)src], y = pts[3 * (size_t)src + 1], z = pts[3 * (size_t)src + 2];
      spts[i] = float4{x, y, z, __uint_as_float(src)};
      const float v[3] = {x, y, z};
#pragma unroll
      for (int a = 0; a < 3; a++) { mn[a] = min(mn[a], lr_ord(v[a])); mx[a] = max(mx[a], lr_ord(v[a])); }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) { atomicMin(&smin[a], mn[a]); atomicMax(&smax[a], mx[a]); }
  __syncthreads();
  if (threadIdx.x < 3) {
    boxes[6 * (size_t)blockIdx.x + threadIdx.x] = lr_unord(smin[threadIdx.x]);
    boxes[6 * (size_t)blockIdx.x + 3 + threadIdx.x] = lr_unord(smax[threadIdx.x]);
  }
}

LR_DEV void lr_best3(float d, float best[3]) {
  if (d < best[2]) {
    if (d < best[1]) {
      best[2] = best[1];
      if (d < best[0]) { best[1] = best[0]; best[0] = d; } else { best[1] = d; }
    } else {
      best[2] = d;
    }
  }
}

__global__ void __launch_bounds__(256)
lr_knn_search_kernel(int P, const float4* __restrict__ spts, const float* __restrict__ boxes, int nboxes,
                     float* __restrict__ out) {
  __shared__ float4 stage[LR_KNN_BOX];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < P;
  float4 me = live ? spts[i] : float4{0.f, 0.f, 0.f, 0.f};
  float best[3] = {3.4e38f, 3.4e38f, 3.4e38f};
  if (live) {
    for (int k = max(0, i - 3); k <= min(P - 1, i + 3); k++) {
      if (k == i) continue;
      const float4 q = spts[k];
      const float dx = q.x - me.x, dy = q.y - me.y, dz = q.z - me.z;
      lr_best3(dx * dx + dy * dy + dz * dz, best);
    }
  }
  const int b_home = (blockIdx.x * 256) / LR_KNN_BOX;  // the box this workgroup's points live in: visit it first
  for (int step = 0; step < nboxes; step++) {
    const int b = step == 0 ? b_home : (step <= b_home ? step - 1 : step);
    const float* bb = boxes + 6 * (size_t)b;  // wave-uniform -> scalar loads
    float d = 0.f;
    {
      const float lx = bb[0], ly = bb[1], lz = bb[2], hx = bb[3], hy = bb[4], hz = bb[5];
      const float ex = fmaxf(fmaxf(lx - me.x, me.x - hx), 0.f);
      const float ey = fmaxf(fmaxf(ly - me.y, me.y - hy), 0.f);
      const float ez = fmaxf(fmaxf(lz - me.z, me.z - hz), 0.f);
      d = ex * ex + ey * ey + ez * ez;
    }
    const bool want = live && !(d > best[2]);
    if (!__syncthreads_or(want ? 1 : 0)) continue;  // nobody in this workgroup can improve from box b
    const int base = b * LR_KNN_BOX, cnt = min(LR_KNN_BOX, P - base);
    for (int k = threadIdx.x; k < cnt; k += 256) stage[k] = spts[base + k];
    __syncthreads();
    if (want) {
      for (int k = 0; k < cnt; k++) {
        if (abs(base + k - i) <= 3) continue;  // itself and the +-3 Morton neighbours already seeded above
        const float4 q = stage[k];
        const float dx = q.x - me.x, dy = q.y - me.y, dz = q.z - me.z;
        lr_best3(dx * dx + dy * dy + dz * dz, best);
      }
    }
    __syncthreads();
  }
  if (live) out[__float_as_uint(me.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

// ---- scratch layout -----------------------------------------------------------------------------------------
struct LrKnnLayout {
  size_t bbox, codes, idx, codes2, idx2, spts, boxes, sort_tmp, total, sort_tmp_bytes;
};
static size_t lr_align(size_t x) { return (x + 255) & ~(size_t)255; }
static hipError_t lr_knn_layout(int P, LrKnnLayout* L) {
  size_t tmp = 0;
  uint32_t* nk = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, nk, nk, nk, nk, (size_t)P, 0, 30, (hipStream_t)0);
  if (e != hipSuccess) return e;
  const size_t nboxes = ((size_t)P + LR_KNN_BOX - 1) / LR_KNN_BOX;
  size_t off = 0;
  L->bbox = off; off += lr_align(6 * sizeof(uint32_t));
  L->codes = off; off += lr_align(sizeof(uint32_t) * (size_t)P);
  L->idx = off; off += lr_align(sizeof(uint32_t) * (size_t)P);
  L->codes2 = off; off += lr_align(sizeof(uint32_t) * (size_t)P);
  L->idx2 = off; off += lr_align(sizeof(uint32_t) * (size_t)P);
  L->spts = off; off += lr_align(sizeof(float4) * (size_t)P);
  L->boxes = off; off += lr_align(sizeof(float) * 6 * nboxes);
  L->sort_tmp = off; off += lr_align(tmp);
  L->sort_tmp_bytes = tmp;
  L->total = off;
  return hipSuccess;
}

size_t lr_knn_scratch_bytes(int P) {
  if (P <= 0) return 0;
  LrKnnLayout L;
  if (lr_knn_layout(P, &L) != hipSuccess) return 0;
  return L.total;
}

hipError_t lr_launch_knn(int P, const float* pts, float* out, void* scratch, size_t scratch_bytes, hipStream_t s) {
  if (P <= 0) return hipSuccess;
  LrKnnLayout L;
  hipError_t e = lr_knn_layout(P, &L);
  if (e != hipSuccess) return e;
  if (scratch_bytes < L.total) return hipErrorInvalidValue;
  char* base = reinterpret_cast<char*>(scratch);
  uint32_t* bbox = reinterpret_cast<uint32_t*>(base + L.bbox);
  uint32_t* codes = reinterpret_cast<uint32_t*>(base + L.codes);
  uint32_t* idx = reinterpret_cast<uint32_t*>(base + L.idx);
  uint32_t* codes2 = reinterpret_cast<uint32_t*>(base + L.codes2);
  uint32_t* idx2 = reinterpret_cast<uint32_t*>(base + L.idx2);
  float4* spts = reinterpret_cast<float4*>(base + L.spts);
  float* boxes = reinterpret_cast<float*>(base + L.boxes);
  const int nboxes = (P + LR_KNN_BOX - 1) / LR_KNN_BOX;
  lr_prof_begin(LRK_MISC, s);
  if ((e = hipMemsetAsync(bbox, 0xff, 3 * sizeof(uint32_t), s)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(bbox + 3, 0x00, 3 * sizeof(uint32_t), s)) != hipSuccess) return e;
  const int g = min((P + 255) / 256, 2048);
  hipLaunchKernelGGL(lr_knn_bbox_kernel, dim3(g), dim3(256), 0, s, P, pts, bbox);
  hipLaunchKernelGGL(lr_knn_morton_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, pts, bbox, codes, idx);
  size_t tmp = L.sort_tmp_bytes;
  e = rocprim::radix_sort_pairs(base + L.sort_tmp, tmp, codes, codes2, idx, idx2, (size_t)P, 0, 30, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(lr_knn_gather_kernel, dim3(nboxes), dim3(256), 0, s, P, pts, idx2, spts, boxes);
  hipLaunchKernelGGL(lr_knn_search_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, spts, boxes, nboxes, out);
  lr_prof_end(LRK_MISC, s);
  return hipGetLastError();
}

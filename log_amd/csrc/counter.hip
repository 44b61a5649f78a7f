// counter.hip -- N4: the bookkeeping LoG runs on the rasterizer's outputs after every view.
//   * lr_launch_id_histogram : torch.unique(point_id_pixel, sorted=True, return_counts=True) minus the leading -1
//                              (/root/reference/LoG/render/renderer.py:156-159) -- the reference sorts all H*W ids;
//                              ids are < n, so a count per Gaussian + an ordered compaction gives the same lists.
//   * lr_launch_counter      : Counter.update_by_output for one view (/root/reference/LoG/model/counter.py:36-68),
//                              ~25 indexing kernels in the reference, one launch here.
//   * lr_launch_sparse_adam  : SparseOptimizer.step (/root/reference/LoG/model/sparse_optimizer.py:41-78,163-196),
//                              all keys in one launch, rows selected by flag_vis in the kernel (no compaction, no
//                              index.cpu(), no state gather/scatter round trip).
// All of it is streaming / scattered-row work: HBM-bound, no LDS tiling needed beyond the block scans.
#include "common.hpp"

#define CNT_CHUNK 1024u
#define CNT_HDR_WORDS 4u   // [0] number of distinct ids

static inline size_t cnt_align4(size_t w) { return (w + 3) & ~(size_t)3; }
size_t lr_hist_scratch_bytes(int n) {
  const size_t nn = (size_t)(n > 0 ? n : 0);
  return 4 * (CNT_HDR_WORDS + cnt_align4(nn) + cnt_align4(nn / CNT_CHUNK + 1));
}

// One pixel per lane, rows of the id map are contiguous: a wave sees runs of equal ids (a splat wins several
// neighbouring pixels), so the head of every run adds the run length -- one atomic per run instead of per pixel.
__global__ void __launch_bounds__(256)
cnt_hist_kernel(const int32_t* __restrict__ pid, uint32_t npix, uint32_t n, uint32_t* __restrict__ count) {
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t base = (blockIdx.x * 256u + (threadIdx.x & ~63u)); base < npix; base += gridDim.x * 256u) {
    const uint32_t i = base + lane;
    const int32_t id = i < npix ? pid[i] : -1;
    const int32_t prev = __shfl_up(id, 1);
    const bool head = lane == 0 || prev != id;
    const uint64_t heads = __ballot(head);
    const uint64_t later = lane == 63u ? 0ull : (heads >> (lane + 1u));
    const uint32_t run = later ? (uint32_t)__builtin_ctzll(later) + 1u : 64u - lane;
    if (head && id >= 0 && (uint32_t)id < n) atomicAdd(&count[id], run);
  }
}

__global__ void __launch_bounds__(256)
cnt_chunk_count_kernel(const uint32_t* __restrict__ count, uint32_t n, uint32_t* __restrict__ chunk_cnt) {
  __shared__ uint32_t ws[4];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t nchunks = (n + CNT_CHUNK - 1) / CNT_CHUNK;
  for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t i = chunk * CNT_CHUNK + (uint32_t)k * 256u + threadIdx.x;
      c += (uint32_t)__popcll(__ballot(i < n && count[i] != 0u));
    }
    if (lane == 0) ws[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) chunk_cnt[chunk] = ws[0] + ws[1] + ws[2] + ws[3];
    __syncthreads();
  }
}

__global__ void __launch_bounds__(1024)
cnt_scan_kernel(uint32_t* __restrict__ chunk_cnt, uint32_t nchunks, uint32_t* __restrict__ hdr) {
  __shared__ uint32_t wk[16];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nchunks; base += 1024u) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t k = i < nchunks ? chunk_cnt[i] : 0u;
    uint32_t ik = k;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t u = __shfl_up(ik, d);
      if ((int)lane >= d) ik += u;
    }
    if (lane == 63u) wk[wave] = ik;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16u; w++) {
      const uint32_t v = wk[w];
      if (w < wave) off += v;
      tot += v;
    }
    if (i < nchunks) chunk_cnt[i] = carry + off + ik - k;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) hdr[0] = carry;
}

__global__ void __launch_bounds__(256)
cnt_emit_kernel(const uint32_t* __restrict__ count, uint32_t n, const uint32_t* __restrict__ chunk_cnt,
                int32_t* __restrict__ ids, int64_t* __restrict__ counts) {
  __shared__ uint32_t cnt[16];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t below = (1ull << lane) - 1ull;
  const uint32_t nchunks = (n + CNT_CHUNK - 1) / CNT_CHUNK;
  for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    uint32_t c[4];
    uint64_t b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t i = chunk * CNT_CHUNK + (uint32_t)k * 256u + threadIdx.x;
      c[k] = i < n ? count[i] : 0u;
      b[k] = __ballot(c[k] != 0u);
      if (lane == 0) cnt[k * 4 + wave] = (uint32_t)__popcll(b[k]);
    }
    __syncthreads();
    uint32_t pre = chunk_cnt[chunk], e = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      for (; e < (uint32_t)k * 4u + wave; e++) pre += cnt[e];
      if (c[k] != 0u) {
        const uint32_t o = pre + (uint32_t)__popcll(b[k] & below);
        ids[o] = (int32_t)(chunk * CNT_CHUNK + (uint32_t)k * 256u + threadIdx.x);
        counts[o] = (int64_t)c[k];
      }
    }
    __syncthreads();
  }
}

hipError_t lr_launch_id_histogram(int n, const int32_t* pid, int npix, int32_t* ids, int64_t* counts, void* scratch,
                                  hipStream_t s) {
  uint32_t* w = reinterpret_cast<uint32_t*>(scratch);
  uint32_t* count = w + CNT_HDR_WORDS;
  uint32_t* chunk_cnt = count + cnt_align4((size_t)n);
  hipError_t e = hipMemsetAsync(w, 0, 4 * (CNT_HDR_WORDS + cnt_align4((size_t)n)), s);
  if (e != hipSuccess) return e;
  lr_prof_begin(LRK_HIST, s);
  const uint32_t nchunks = ((uint32_t)n + CNT_CHUNK - 1) / CNT_CHUNK;
  if (npix > 0 && n > 0) {
    const uint32_t g = ((uint32_t)npix + 255u) / 256u;
    hipLaunchKernelGGL(cnt_hist_kernel, dim3(g > 4096u ? 4096u : g), dim3(256), 0, s, pid, (uint32_t)npix, (uint32_t)n, count);
  }
  if (nchunks) {
    const uint32_t g = nchunks > 2048u ? 2048u : nchunks;
    hipLaunchKernelGGL(cnt_chunk_count_kernel, dim3(g), dim3(256), 0, s, (const uint32_t*)count, (uint32_t)n, chunk_cnt);
    hipLaunchKernelGGL(cnt_scan_kernel, dim3(1), dim3(1024), 0, s, chunk_cnt, nchunks, w);
    hipLaunchKernelGGL(cnt_emit_kernel, dim3(g), dim3(256), 0, s, (const uint32_t*)count, (uint32_t)n,
                       (const uint32_t*)chunk_cnt, ids, counts);
  }
  lr_prof_end(LRK_HIST, s);
  return hipGetLastError();
}

// ---- Counter.update_by_output, one view (counter.py:36-68) --------------------------------------------------------
// Threads [0, nv): the "seen this view" statistics of every submitted Gaussian with radii > 0 (:48-50, :58-62, :67).
// Threads [nv, nv + k): the "won pixels" statistics of the k distinct ids of point_id_pixel (:52-57, :63-68).
// visible_index has no duplicates and the two groups write different buffers, so no atomics are needed.

__global__ void __launch_bounds__(256)
cnt_update_kernel(CounterArgs a) {
  const int32_t t = (int32_t)(blockIdx.x * 256u + threadIdx.x);
  if (t < a.nv) {
    const int32_t r = a.radii[t];
    const bool vis = r > 0;
    if (a.flag_vis) a.flag_vis[t] = vis ? 1 : 0;
    if (!vis) return;
    const int64_t row = a.visible_index[t];
    if (row < 0 || row >= a.num_points) return;
    const float w = a.weight[t];
    a.create_steps[row] += 1;
    a.visible_count[row] = (int16_t)(a.visible_count[row] + 1);
    a.weights_max[row] = fmaxf(a.weights_max[row], w);
    a.weights_sum[row] = a.weights_sum[row] + w;
    const int16_t rs = (int16_t)r;                      // radii.short() (:67)
    const int16_t old = a.radii_max[row];
    a.radii_max[row] = old > rs ? old : rs;
  } else if (t < a.nv + a.k) {
    const int32_t j = t - a.nv;
    const int32_t id = a.point_id[j];
    if (id < 0 || id >= a.nv) return;
    const int64_t row = a.visible_index[id];
    if (row < 0 || row >= a.num_points) return;
    const int64_t c = a.point_count[j];
    a.area_sum[row] = (int32_t)((int64_t)a.area_sum[row] + c);
    const float gx = a.grad[3 * (size_t)id], gy = a.grad[3 * (size_t)id + 1];
    const float gn = sqrtf(lr_fma(gx, gx, gy * gy));    // torch.norm(grad[:, :2], dim=-1) (:46)
    a.grad_sum[row] = a.grad_sum[row] + gn * (float)c;
    const int32_t ci = (int32_t)c;
    const int32_t old = a.radii_max_max[row];
    a.radii_max_max[row] = ci > old ? ci : old;
  }
}

hipError_t lr_launch_counter(const CounterArgs& a, hipStream_t s) {
  const int64_t total = (int64_t)a.nv + (int64_t)a.k;
  if (total <= 0) return hipSuccess;
  lr_prof_begin(LRK_COUNTER, s);
  hipLaunchKernelGGL(cnt_update_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, a);
  lr_prof_end(LRK_COUNTER, s);
  return hipGetLastError();
}

// ---- SparseOptimizer.step (sparse_optimizer.py:41-78,163-196) -----------------------------------------------------
// One thread per (submitted row, element) of one key (blockIdx.y); rows with flag_vis == 0 are skipped.  The op
// sequence of _single_tensor_adam, fp32:  m = fma(g, 1-b1, m*b1); v = fma((1-b2)*g, g, v*b2);
// denom = sqrt(amsgrad ? max(vmax, v) : v) / sqrt(bias_correction2) + eps;  p = p + (-step_size) * (m / denom).

// ADAM_UNROLL elements per thread and round, a grid apart (each a coalesced access of its wave), their flag / index
// reads and then all their moments and gradients requested together.  Measured on 6.3 M of 13 M rows x 59 floats
// (tools/adam_probe.py, profiles/r03_probes.md; algorithmic GB/s at 28 B per element), unroll 1 / 4:  contiguous rows
// 5070 / 4640, groups of 16 rows 4930 / 4060, groups of 4 (siblings of a 4-ary tree) 4180 / 3330, random rows 3280 / 2570
// -- more requests in flight only spread the partial lines of neighbouring rows further apart in time; four ADJACENT
// elements per thread (16-byte strides per lane) lose more (3010 contiguous).  So 1 it is.  IDX32: element indices fit
// 31 bits -- 32-bit instead of 64-bit division for (row, column): 4280 -> 5070 contiguous, 2950 -> 3280 random (a
// division-free variant had changed nothing in round 2 -- on rows gathered through a 64-bit index all the same).
#define ADAM_UNROLL 1
template <bool IDX32>
__global__ void __launch_bounds__(256)
adam_kernel(AdamArgs a) {
  const AdamKey& k = a.key[blockIdx.y];
  const int64_t total = (int64_t)a.m * k.width;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x; e0 < total; e0 += ADAM_UNROLL * stride) {
    int64_t e[ADAM_UNROLL], row[ADAM_UNROLL];
    int32_t c[ADAM_UNROLL];
    bool ok[ADAM_UNROLL];
#pragma unroll
    for (int u = 0; u < ADAM_UNROLL; u++) {
      e[u] = e0 + u * stride;
      ok[u] = e[u] < total;
      row[u] = -1;
      c[u] = 0;
      if (ok[u]) {
        int32_t r;
        if (IDX32) { r = (int32_t)((uint32_t)e[u] / (uint32_t)k.width); c[u] = (int32_t)((uint32_t)e[u] - (uint32_t)r * (uint32_t)k.width); }
        else { r = (int32_t)(e[u] / k.width); c[u] = (int32_t)(e[u] - (int64_t)r * k.width); }
        if (a.flag_vis[r]) row[u] = a.index[r];
      }
    }
    float g[ADAM_UNROLL], m0[ADAM_UNROLL], v0[ADAM_UNROLL], p0[ADAM_UNROLL], vm[ADAM_UNROLL];
    size_t o[ADAM_UNROLL];
#pragma unroll
    for (int u = 0; u < ADAM_UNROLL; u++) {
      ok[u] = ok[u] && row[u] >= 0 && row[u] < a.num_points;
      o[u] = 0; g[u] = 0.f; m0[u] = 0.f; v0[u] = 0.f; p0[u] = 0.f; vm[u] = 0.f;
      if (ok[u]) {
        o[u] = (size_t)row[u] * (size_t)k.width + (size_t)c[u];
        g[u] = k.grad[e[u]];
        m0[u] = k.exp_avg[o[u]];
        v0[u] = k.exp_avg_sq[o[u]];
        p0[u] = k.param[e[u]];
        if (k.max_exp_avg_sq) vm[u] = k.max_exp_avg_sq[o[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < ADAM_UNROLL; u++) {
      if (!ok[u]) continue;
      const float m = lr_fma(g[u], a.omb1, m0[u] * a.beta1);
      const float v = lr_fma(a.omb2 * g[u], g[u], v0[u] * a.beta2);
      k.exp_avg[o[u]] = m;
      k.exp_avg_sq[o[u]] = v;
      float vd = v;
      if (k.max_exp_avg_sq) {
        vd = fmaxf(vm[u], v);
        k.max_exp_avg_sq[o[u]] = vd;
      }
      const float denom = sqrtf(vd) / a.bc2_sqrt + a.eps;
      k.model[o[u]] = p0[u] + k.neg_step_size * (m / denom);
    }
  }
}

hipError_t lr_launch_sparse_adam(const AdamArgs& a, int num_keys, hipStream_t s) {
  if (a.m <= 0 || num_keys <= 0) return hipSuccess;
  int maxw = 1;
  for (int i = 0; i < num_keys; i++) maxw = a.key[i].width > maxw ? a.key[i].width : maxw;
  int64_t blocks = ((int64_t)a.m * maxw + 256 * ADAM_UNROLL - 1) / (256 * ADAM_UNROLL);
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  lr_prof_begin(LRK_ADAM, s);
  if ((int64_t)a.m * maxw < (int64_t)0x7fffffff)
    hipLaunchKernelGGL(adam_kernel<true>, dim3((uint32_t)blocks, (uint32_t)num_keys), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(adam_kernel<false>, dim3((uint32_t)blocks, (uint32_t)num_keys), dim3(256), 0, s, a);
  lr_prof_end(LRK_ADAM, s);
  return hipGetLastError();
}

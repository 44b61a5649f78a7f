// lod.hip -- N3: level-of-detail selection, i.e. TensorTree.traverse + _query_tree_torch
// (/root/reference/LoG/model/tensor_tree.py:131-185) with Gaussian.compute_radius
// (/root/reference/LoG/model/level_of_gaussian.py:65-88: gather -> exp / normalize -> compute_radius) fused into
// the per-level pass.  The reference walks the tree level by level with ~15 torch kernels and several host syncs per
// level (boolean-mask indexing, `.sum() == 0`); here a level is three launches whose sizes are read from device
// memory, so the whole selection costs ONE host sync (the final count), and the output order is the reference's:
// [kept roots | kept children of level 1 | ... | the frontier left when the depth limit is reached], each group
// in the order of its parents (stable compaction).
//
// Per level: classify (one thread per child slot: follow frontier -> node -> child, gather the child's 40 bytes,
// activate, project, decide keep/next; per-1024-slot counts) -> scan of the chunk counts (one workgroup) ->
// scatter (ballot ranks inside the chunk + the chunk's prefix).  A point is visited at most once per call.
#include "common.hpp"

#define LOD_MAX_LEVELS 128
#define LOD_HDR_COUNT 0                          // [l] frontier size entering level l (l >= 1)
#define LOD_HDR_OUT (LOD_MAX_LEVELS + 4)         // [l] output cursor before level l's keeps
#define LOD_HDR_TOTAL (2 * (LOD_MAX_LEVELS + 4))
#define LOD_HDR_OVERFLOW (LOD_HDR_TOTAL + 1)
#define LOD_HDR_LEFT (LOD_HDR_TOTAL + 2)       // size of the frontier that was appended unexpanded at the depth limit
#define LOD_HDR_WORDS (LOD_HDR_TOTAL + 8)
#define LOD_CHUNK 1024u
#define LOD_NONE 0xFFFFFFFFu
#define LOD_NEXT 0x80000000u

struct LodArgs {
  const int32_t* node_index;
  const int32_t* tree;
  const float* xyz;
  const float* scaling;
  const float* rotation;
  const int64_t* root_index;
  const float* proj;
  const float* view;
  float fx, fy, tanfovx, tanfovy, min_px;
  int32_t num_points, num_nodes, max_child, num_roots;
  uint32_t* hdr;
  uint32_t* code;
  uint32_t* chunk_keep;
  uint32_t* chunk_next;
  int64_t* out;
  uint32_t out_capacity, frontier_capacity;
};

static inline size_t lod_align4(size_t w) { return (w + 3) & ~(size_t)3; }
static inline size_t lod_slots(int num_roots, int num_nodes, int max_child) {
  size_t a = (size_t)(num_roots > 0 ? num_roots : 0), b = (size_t)(num_nodes > 0 ? num_nodes : 0) * (size_t)max_child;
  return a > b ? a : b;
}
size_t lr_lod_scratch_bytes(int num_roots, int num_nodes, int max_child) {
  const size_t slots = lod_slots(num_roots, num_nodes, max_child);
  const size_t chunks = slots / LOD_CHUNK + 1;
  return 4 * (LOD_HDR_WORDS + 2 * lod_align4((size_t)(num_nodes > 0 ? num_nodes : 0)) + lod_align4(slots) + 2 * lod_align4(chunks));
}

template <bool ROOTS>
LR_DEV uint32_t lod_num_slots(const LodArgs& a, int level) {
  return ROOTS ? (uint32_t)a.num_roots : a.hdr[LOD_HDR_COUNT + level] * (uint32_t)a.max_child;
}

// keep = (projected radius < min_resolution_pixel) | is_leaf   (tensor_tree.py:143-146, :169-171)
template <bool ROOTS>
__global__ void __launch_bounds__(256)
lod_classify_kernel(LodArgs a, int level, const uint32_t* __restrict__ frontier) {
  __shared__ uint32_t wsum[2][4];
  const uint32_t n_slots = lod_num_slots<ROOTS>(a, level);
  const uint32_t nchunks = (n_slots + LOD_CHUNK - 1) / LOD_CHUNK;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t mc = (uint32_t)a.max_child;
  for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    uint32_t nk = 0, nn = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t slot = chunk * LOD_CHUNK + (uint32_t)k * 256u + threadIdx.x;
      int32_t c = -1;
      if (slot < n_slots) {
        if (ROOTS) {
          c = (int32_t)a.root_index[slot];
        } else {
          const uint32_t f = slot / mc, j = slot - f * mc;
          const int32_t node = a.node_index[frontier[f]];     // frontier entries always have children
          c = (node >= 0 && node < a.num_nodes) ? a.tree[(size_t)node * mc + j] : -1;
        }
      }
      const bool valid = c >= 0 && c < a.num_points;
      uint32_t cls = 0;
      if (valid) {
        bool keep = a.node_index[c] == -1;
        if (!keep) {
          const float p[3] = {a.xyz[3 * (size_t)c], a.xyz[3 * (size_t)c + 1], a.xyz[3 * (size_t)c + 2]};
          const float s[3] = {lr_exp_any(a.scaling[3 * (size_t)c]), lr_exp_any(a.scaling[3 * (size_t)c + 1]),
                              lr_exp_any(a.scaling[3 * (size_t)c + 2])};
          const float4 q4 = reinterpret_cast<const float4*>(a.rotation)[c];
          float q[4] = {q4.x, q4.y, q4.z, q4.w};
          lr_normalize4(q);
          keep = lr_radius_one(p, s, q, a.proj, a.view, a.fx, a.fy, a.tanfovx, a.tanfovy) < a.min_px;
        }
        cls = keep ? 1u : 2u;
      }
      if (slot < n_slots) a.code[slot] = valid ? ((uint32_t)c | (cls == 2u ? LOD_NEXT : 0u)) : LOD_NONE;
      nk += (uint32_t)__popcll(__ballot(cls == 1u));
      nn += (uint32_t)__popcll(__ballot(cls == 2u));
    }
    if (lane == 0) { wsum[0][wave] = nk; wsum[1][wave] = nn; }
    __syncthreads();
    if (threadIdx.x == 0) {
      a.chunk_keep[chunk] = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
      a.chunk_next[chunk] = wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
    }
    __syncthreads();
  }
}

// exclusive prefixes of the chunk counts (keep: offset by the output cursor), next level's header entries
template <bool ROOTS>
__global__ void __launch_bounds__(1024)
lod_scan_kernel(LodArgs a, int level) {
  __shared__ uint32_t wk[16], wn[16];
  const uint32_t n_slots = lod_num_slots<ROOTS>(a, level);
  const uint32_t nchunks = (n_slots + LOD_CHUNK - 1) / LOD_CHUNK;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t ck = a.hdr[LOD_HDR_OUT + level], cn = 0;
  for (uint32_t base = 0; base < nchunks; base += 1024u) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t k = i < nchunks ? a.chunk_keep[i] : 0u, n = i < nchunks ? a.chunk_next[i] : 0u;
    uint32_t ik = k, in_ = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t uk = __shfl_up(ik, d), un = __shfl_up(in_, d);
      if ((int)lane >= d) { ik += uk; in_ += un; }
    }
    if (lane == 63u) { wk[wave] = ik; wn[wave] = in_; }
    __syncthreads();
    uint32_t ok = 0, on = 0, tk = 0, tn = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16u; w++) {
      const uint32_t vk = wk[w], vn = wn[w];
      if (w < wave) { ok += vk; on += vn; }
      tk += vk; tn += vn;
    }
    if (i < nchunks) {
      a.chunk_keep[i] = ck + ok + ik - k;
      a.chunk_next[i] = cn + on + in_ - n;
    }
    ck += tk; cn += tn;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    a.hdr[LOD_HDR_OUT + level + 1] = ck;
    a.hdr[LOD_HDR_COUNT + level + 1] = cn;
  }
}

template <bool ROOTS>
__global__ void __launch_bounds__(256)
lod_scatter_kernel(LodArgs a, int level, uint32_t* __restrict__ next_frontier) {
  __shared__ uint32_t cnt[2][16];  // [class][k * 4 + wave]: slot order inside a chunk is (k, wave, lane)
  const uint32_t n_slots = lod_num_slots<ROOTS>(a, level);
  const uint32_t nchunks = (n_slots + LOD_CHUNK - 1) / LOD_CHUNK;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t below = (1ull << lane) - 1ull;
  for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    uint32_t code[4];
    uint64_t bk[4], bn[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t slot = chunk * LOD_CHUNK + (uint32_t)k * 256u + threadIdx.x;
      code[k] = slot < n_slots ? a.code[slot] : LOD_NONE;
      const bool valid = code[k] != LOD_NONE, nxt = valid && (code[k] & LOD_NEXT);
      bk[k] = __ballot(valid && !nxt);
      bn[k] = __ballot(nxt);
      if (lane == 0) { cnt[0][k * 4 + wave] = (uint32_t)__popcll(bk[k]); cnt[1][k * 4 + wave] = (uint32_t)__popcll(bn[k]); }
    }
    __syncthreads();
    const uint32_t kb = a.chunk_keep[chunk], nb = a.chunk_next[chunk];
    uint32_t pk = 0, pn = 0;   // keeps / nexts of this chunk in front of (k, wave)
    uint32_t e = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      for (; e < (uint32_t)k * 4u + wave; e++) { pk += cnt[0][e]; pn += cnt[1][e]; }
      const uint32_t id = code[k] & ~LOD_NEXT;
      if (code[k] != LOD_NONE) {
        if (code[k] & LOD_NEXT) {
          const uint32_t idx = nb + pn + (uint32_t)__popcll(bn[k] & below);
          if (idx < a.frontier_capacity) next_frontier[idx] = id;
          else a.hdr[LOD_HDR_OVERFLOW] = 1u;
        } else {
          const uint32_t idx = kb + pk + (uint32_t)__popcll(bk[k] & below);
          if (idx < a.out_capacity) a.out[idx] = (int64_t)id;
          else a.hdr[LOD_HDR_OVERFLOW] = 1u;
        }
      }
    }
    __syncthreads();
  }
}

// depth limit reached (tensor_tree.py:134-137): whatever is still on the frontier is taken as it is
__global__ void __launch_bounds__(256)
lod_finish_kernel(LodArgs a, int level, const uint32_t* __restrict__ frontier) {
  const uint32_t n = a.hdr[LOD_HDR_COUNT + level], base = a.hdr[LOD_HDR_OUT + level];
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    if (base + i < a.out_capacity) a.out[base + i] = (int64_t)frontier[i];
    else a.hdr[LOD_HDR_OVERFLOW] = 1u;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { a.hdr[LOD_HDR_TOTAL] = base + n; a.hdr[LOD_HDR_LEFT] = n; }
}

hipError_t lr_launch_lod(int num_points, int num_nodes, int max_child, const int32_t* node_index, const int32_t* tree,
                         const float* xyz, const float* scaling, const float* rotation, const int64_t* root_index,
                         int num_roots, const float* proj, const float* view, float fx, float fy, float tanfovx,
                         float tanfovy, float min_px, int levels, int64_t* out, uint32_t out_capacity, void* scratch,
                         hipStream_t s) {
  uint32_t* w = reinterpret_cast<uint32_t*>(scratch);
  const size_t nn = lod_align4((size_t)(num_nodes > 0 ? num_nodes : 0));
  const size_t slots = lod_slots(num_roots, num_nodes, max_child);
  const size_t chunks = slots / LOD_CHUNK + 1;
  LodArgs a;
  a.node_index = node_index; a.tree = tree; a.xyz = xyz; a.scaling = scaling; a.rotation = rotation;
  a.root_index = root_index; a.proj = proj; a.view = view;
  a.fx = fx; a.fy = fy; a.tanfovx = tanfovx; a.tanfovy = tanfovy; a.min_px = min_px;
  a.num_points = num_points; a.num_nodes = num_nodes; a.max_child = max_child; a.num_roots = num_roots;
  a.hdr = w;
  uint32_t* front[2] = {w + LOD_HDR_WORDS, w + LOD_HDR_WORDS + nn};
  a.code = w + LOD_HDR_WORDS + 2 * nn;
  a.chunk_keep = a.code + lod_align4(slots);
  a.chunk_next = a.chunk_keep + lod_align4(chunks);
  a.out = out; a.out_capacity = out_capacity; a.frontier_capacity = (uint32_t)(num_nodes > 0 ? num_nodes : 0);
  hipError_t e = hipMemsetAsync(w, 0, 4 * LOD_HDR_WORDS, s);
  if (e != hipSuccess) return e;
  lr_prof_begin(LRK_LOD, s);
  const uint32_t root_chunks = ((uint32_t)num_roots + LOD_CHUNK - 1) / LOD_CHUNK;
  if (root_chunks) hipLaunchKernelGGL(lod_classify_kernel<true>, dim3(root_chunks), dim3(256), 0, s, a, 0, (const uint32_t*)nullptr);
  hipLaunchKernelGGL(lod_scan_kernel<true>, dim3(1), dim3(1024), 0, s, a, 0);
  if (root_chunks) hipLaunchKernelGGL(lod_scatter_kernel<true>, dim3(root_chunks), dim3(256), 0, s, a, 0, front[0]);
  // Below the roots the sizes live on the device: a fixed grid strides over however many chunks the level has.
  size_t level_chunks = ((size_t)(num_nodes > 0 ? num_nodes : 0) * (size_t)max_child + LOD_CHUNK - 1) / LOD_CHUNK;
  const uint32_t grid = (uint32_t)(level_chunks < 1 ? 1 : (level_chunks > 2048 ? 2048 : level_chunks));
  int cur = 0;
  for (int level = 1; level <= levels; level++) {
    hipLaunchKernelGGL(lod_classify_kernel<false>, dim3(grid), dim3(256), 0, s, a, level, (const uint32_t*)front[cur]);
    hipLaunchKernelGGL(lod_scan_kernel<false>, dim3(1), dim3(1024), 0, s, a, level);
    hipLaunchKernelGGL(lod_scatter_kernel<false>, dim3(grid), dim3(256), 0, s, a, level, front[cur ^ 1]);
    cur ^= 1;
  }
  hipLaunchKernelGGL(lod_finish_kernel, dim3(grid > 256 ? 256 : grid), dim3(256), 0, s, a, levels + 1, (const uint32_t*)front[cur]);
  lr_prof_end(LRK_LOD, s);
  return hipGetLastError();
}

int lr_lod_max_levels() { return LOD_MAX_LEVELS; }
uint32_t lr_lod_total_word() { return LOD_HDR_TOTAL; }

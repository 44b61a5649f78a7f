// api.hip -- extern "C" entry points of liblograst.so (declared in include/lograst.h), launch
// sequencing, error text and the HIP-event per-kernel timer.  No allocation, no host sync except where
// the header says so.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <atomic>
#include <string>
#include <vector>

#include "common.hpp"

// launchers (defined next to their kernels)
void lr_launch_radius(int P, const float* means, const float* scales, const float* rots, const float* proj,
                      const float* view, float fx, float fy, float tanfovx, float tanfovy, float* radii,
                      hipStream_t s);
void lr_launch_project(const LrView& v, int N, const float* means, const float* scales, const float* rots,
                       const float* opac, const float* colors, int* radii, void* geom, uint32_t* ranked,
                       uint32_t* big, uint32_t* hdr, uint32_t* basetab, int batch, int planes, int tile_cull,
                       hipStream_t s);
void lr_launch_scan(uint32_t* state, uint32_t tiles, uint32_t cs, uint32_t big_off, hipStream_t s);
void lr_launch_rebase(uint32_t* state, uint32_t tiles, uint32_t batches, uint32_t t_lo, uint32_t t_hi, hipStream_t s);
bool lr_band_sparse(const LrView& v, int batch);
void lr_launch_zero_words(uint32_t* p, size_t words, hipStream_t s);
void lr_launch_zero_floats(float* p, size_t n, hipStream_t s);
// exchange.hip
void lx_launch_pack_rows(float* rows, int groups, long long rows_per_group, int kmax, float* packed,
                         size_t seg_floats, uint32_t* overflow, int clear, const uint32_t* hint, long long hint_rows,
                         hipStream_t s);
void lx_launch_add_visible(float* seen, const int32_t* radii, long long n, hipStream_t s);
void lx_launch_add_visible_n(float* seen, const int32_t* const* radii, int k, long long n, hipStream_t s);
void lx_launch_unpack_rows(float* dest, const float* packed, int segments, int kmax, size_t seg_floats,
                           long long rows_per_group, long long dest_group_rows, int add, int zero, hipStream_t s);
void lr_launch_fill(int N, int gx, const void* geom, uint32_t* state, uint32_t tiles, uint64_t* keys,
                    uint32_t capacity, uint32_t max_len_hint, uint32_t* status, float* zero_n, float* zero_block,
                    int zero_block_floats, int rebased, int speculative, int band, int staged_k, hipStream_t s);
void lr_launch_tile_rows(const LrView& v, int N, const float* means, const float* scales, const float* rots,
                         uint32_t* rows, hipStream_t s);
void lr_launch_stream_copy(const void* src, void* dst, size_t bytes, int blocks, hipStream_t s);
int lr_launch_sort(uint32_t* state, uint32_t tiles, uint64_t* keys, uint32_t* plist, uint32_t capacity,
                   uint32_t max_len, int lazy, hipStream_t s);
void lr_launch_sort_rest(uint32_t* state, uint32_t tiles, uint64_t* keys, uint32_t* plist, uint32_t capacity,
                         uint32_t max_len, int mode, hipStream_t s);
void lr_launch_ordered_lengths(const uint32_t* state, uint32_t tiles, uint32_t* out, hipStream_t s);
void lr_launch_blend_fwd(const LrView& v, const void* geom, const uint32_t* state, uint32_t tiles,
                         const uint32_t* plist, uint32_t capacity, float* image, float* final_T, int* n_contrib,
                         int* pid, float* pwp, float* pw, float* zero_conic, int big_input, int lazy, uint64_t* masks,
                         hipStream_t s);
int lr_blend_fwd_form(const LrView& v);
void lr_launch_blend_bwd(const LrView& v, const void* geom, const uint32_t* state, uint32_t tiles,
                         const uint32_t* plist, uint32_t capacity, const float* final_T, const int* n_contrib,
                         const float* dL_dimage, float* acc_rows, int big_input, const uint64_t* masks, hipStream_t s);
void lr_launch_project_bwd(const LrView& v, int N, const float* means, const float* scales, const float* rots,
                           const int* radii, const float* g_mean2d, const float* g_conic, const float* rows,
                           float* o_mean2d, float* o_opac, float* o_col, const float* pw,
                           float* g_means3d, float* g_scales, float* g_rots, bool accumulate, bool sink_rows,
                           hipStream_t s);

size_t lr_knn_scratch_bytes(int P);
hipError_t lr_launch_knn(int P, const float* pts, float* out, void* scratch, size_t scratch_bytes, hipStream_t s);

void lr_launch_sh_fwd(int N, int deg, int M, const float* means, const float* campos, const float* shs, float* colors,
                      uint8_t* clamped, hipStream_t s);
void lr_launch_sh_bwd(int N, int deg, int M, const float* means, const float* campos, const float* shs,
                      const uint8_t* clamped, const float* g_colors, float* g_shs, float* g_means, bool accumulate,
                      hipStream_t s);

size_t lr_lod_scratch_bytes(int num_roots, int num_nodes, int max_child);
hipError_t lr_launch_lod(int num_points, int num_nodes, int max_child, const int32_t* node_index, const int32_t* tree,
                         const float* xyz, const float* scaling, const float* rotation, const int64_t* root_index,
                         int num_roots, const float* proj, const float* view, float fx, float fy, float tanfovx,
                         float tanfovy, float min_px, int levels, int64_t* out, uint32_t out_capacity, void* scratch,
                         hipStream_t s);
int lr_lod_max_levels();
uint32_t lr_lod_total_word();   // header words TOTAL, OVERFLOW, LEFT are consecutive

size_t lr_hist_scratch_bytes(int n);
hipError_t lr_launch_id_histogram(int n, const int32_t* pid, int npix, int32_t* ids, int64_t* counts, void* scratch,
                                  hipStream_t s);
hipError_t lr_launch_counter(const CounterArgs& a, hipStream_t s);
hipError_t lr_launch_sparse_adam(const AdamArgs& a, int num_keys, hipStream_t s);

hipError_t lr_launch_gather_activate(const GatherArgs& a, hipStream_t s);
hipError_t lr_launch_activate_bwd(const ActBwdArgs& a, hipStream_t s);
hipError_t lr_launch_activate_bwd_adam(const ActBwdArgs& a, const AdamArgs& f, const float* g_a_xyz, const int32_t* radii,
                                       hipStream_t s);

static thread_local std::string g_err;
static int lr_fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define LR_HIP(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess)                                                                              \
      return lr_fail(LOGRAST_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));              \
  } while (0)

int lr_env_int(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return (e && *e) ? std::atoi(e) : dflt;
}
// ---- performance knobs ------------------------------------------------------------------------------------
// The tunable ones (lograst_knob_info enumerates them for log_amd.tune): name = the environment variable, default, range.
struct LrKnobInfo { const char* name; int dflt, lo, hi; const char* what; };
static const LrKnobInfo kKnobs[] = {
    {"LOGRAST_HELPER_MIN_N", 4000000, 0, 2000000000, "Gaussians from which the helper passes (absolute slot table, touched-only dL/dconic clearing, separate zero-fill kernels) pay for their launches"},
    {"LOGRAST_HIT_MASKS", 1, 0, 1, "the compositing kernels leave their per-chunk support ballots in lograst_view.hit_masks (when the caller provides it) and the reverse walk reads them instead of running the tests again; 0 = ignore the buffer"},
    {"LOGRAST_LAZY_SORT", 1, 0, 1, "lists of more than 4096 keys are ordered over their first window (7680 positions) only; tiles whose walk needs more are marked by the compositing kernels and finished by a second, normally idle sort + compositing pair; 0 = every list to its end up front"},
    {"LOGRAST_PBWD_LIST", 1, 0, 2, "large inputs with running-sum gradients: the chain rule runs over a compact list of the rows with point_weight > 0 (a streaming compaction pass + a list pass) instead of one kernel that tests every row: 0 never, 1 on band views, 2 always"},
    {"LOGRAST_MID_RANK", 1, 0, 1, "rects of 5..16 tiles are RANKED by the batched projection (LDS atomics; 32-byte rank rows in geom), so the fill places them without cursor atomics or support tests; 0 = counted only, placed through the per-tile cursors"},
    {"LOGRAST_MID_COOP", 16, 0, 64, "rects of 5..16 tiles are counted (projection: in waves that hold at most this many of them) and placed (fill: any non-zero value) by the whole wave, four rects per pass, instead of by their lane; 0 = per lane"},
    {"LOGRAST_DEFER_TILES", LR_COOP_TILES, 4, 4096, "rects above this many tiles are counted by lr_count_huge_kernel (one wave per rect) instead of by their lane"},
    {"LOGRAST_HUGE_CHUNK", LR_HUGE_CHUNK, 256, 8192, "Gaussians per workgroup of lr_count_huge_kernel (multiple of 256)"},
    {"LOGRAST_BATCH_PLANES", 4, 1, 4, "consecutive projection batches one workgroup owns"},
    {"LOGRAST_BATCH_SLOTS", 256, 64, 1024, "workgroups per round the batched projection sizes its batches for"},
    {"LOGRAST_SEPARATE_ZERO", 1, 0, 1, "large inputs: zero-fills streamed by kernels of their own instead of inside the fill kernel"},
    {"LOGRAST_FILL_XCD_ORDER", 1, 0, 1, "fill kernel walks the Gaussians XCD-contiguously"},
    {"LOGRAST_FILL_NT", 1, 0, 1, "fill kernel: non-temporal streams for the fill records and zero-fills"},
    {"LOGRAST_XCD_MODE", 3, 0, 3, "blockIdx -> tile mapping of the compositing kernels (3 = longest list first)"},
    {"LOGRAST_PROJECT_BLOCKS", 512, 64, 65536, "grid cap of the unbatched projection kernel"},
    {"LOGRAST_BWD_ROWS", 2, 0, 2, "reverse walk: 1 = row-split form (four 4x4 blocks per wave), 0 = one quadrant per wave, 2 = the view's walk_form hint (none: row-split from LOGRAST_HELPER_MIN_N Gaussians)"},
    {"LOGRAST_FWD_ROWS", 2, 0, 2, "compositing: 1 = row-split form (four 4x4 blocks per wave), 0 = one quadrant per wave, 2 = the view's walk_form hint"},
    {"LOGRAST_FILL_STAGED", 2, 0, 3, "bucket fill of batched full views: K = the batch's slot-table row staged in LDS by workgroups of up to K x 1024 consecutive Gaussians (K per thread), 0 = one table look-up per tile instance"},
    {"LOGRAST_FILL_PER_THREAD", 1, 1, 4, "bucket fill: Gaussians per thread (their fill records are requested together): 1, 2 or 4"},
    {"LOGRAST_BAND_SPARSE", 1, 0, 1, "band views (tile_row_begin/end a proper part of the grid): 1 = the band projection (Gaussians without a rect cost 44 bytes, survivors compacted into full waves), 0 = the full-view kernel"},
    {"LOGRAST_FWD_BLOCK_TEST", 1, 0, 1, "row-split compositing: 1 = exact support test per 4x4 block, 0 = exact for the quadrant + bounding box per block (the reverse walk on the forward's hit masks visits what the forward's test kept)"},
    {"LOGRAST_BWD_BLOCK_TEST", 1, 0, 1, "row-split reverse walk: 1 = exact support test per 4x4 block, 0 = exact for the quadrant + bounding box per block"},
};
static const int kNumKnobs = (int)(sizeof(kKnobs) / sizeof(kKnobs[0]));
static std::mutex g_knob_mu;
static std::vector<std::pair<std::string, int>> g_knob_over;   // overrides set through lograst_set_knob
static std::atomic<unsigned> g_knob_gen{1};
unsigned lr_knob_generation() { return g_knob_gen.load(std::memory_order_acquire); }
int lr_knob_lookup(const char* name, int dflt) {
  {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    for (auto& kv : g_knob_over)
      if (kv.first == name) return kv.second;
  }
  return lr_env_int(name, dflt);
}
// Support cull in the binning kernels (project.hip): on unless LOGRAST_TILE_CULL=0 or lograst_set_tile_cull(0).
static std::atomic<int> g_tile_cull{-1};
// Gaussians per projection batch (project.hip: lr_project_batched_kernel), 0 = unbatched kernel, and the number of
// consecutive batches one workgroup owns (`planes`: one plane of LDS tile counters each).  A batch is big enough that it
// puts several instances into a tile (that is what it saves in memory-side atomics) and at most 32768 Gaussians (16-bit
// ranks); a workgroup takes as many batches as its LDS holds (up to 4), which makes the runs it reserves in a tile
// adjacent (longer contiguous key writes in the fill) and leaves one workgroup per CU per round.
// LOGRAST_BATCH overrides the batch size (0 disables batching), LOGRAST_BATCH_PLANES caps the planes.
struct LrBatching { uint32_t batch, planes; };
static LrBatching lr_pick_batch(int32_t n, uint32_t tiles, uint32_t gx, uint32_t gy) {
  static const int forced = LR_EXPERIMENT_INT("LOGRAST_BATCH", -1);   // experiment builds: Gaussians per batch, 0 = unbatched kernel
  LR_KNOB(max_planes_k, "LOGRAST_BATCH_PLANES", 4);
  const uint32_t max_planes = (uint32_t)max_planes_k;
  if (n <= 0 || tiles > LR_BATCH_MAX_TILES || gx > 8191u || gy > 8191u || forced == 0) return {0u, 1u};  // 13-bit tile coordinates in the fill record
  uint32_t smax = (uint32_t)(LR_BATCH_LDS_BYTES / (sizeof(uint32_t) * (size_t)tiles));
  if (smax > max_planes) smax = max_planes;
  if (smax > 4u) smax = 4u;
  if (smax < 1u) smax = 1u;
  if (forced > 0) {
    const uint32_t b = (uint32_t)((forced > 32768 ? 32768 : forced) + 1023) / 1024u * 1024u;  // 16-bit LDS counts
    return {b, smax};
  }
  // One workgroup of 1024 threads per CU (82 VGPRs): 256 run at a time.  Size the work so that the workgroups fill
  // whole rounds of 256 (10 M Gaussians: 306 batches of 32768 = 1.2 rounds ran as long as 2).
  LR_KNOB(slots_k, "LOGRAST_BATCH_SLOTS", 256);
  const uint32_t slots = (uint32_t)(slots_k > 0 ? slots_k : 256);
  const uint64_t per_round = (uint64_t)slots * 32768u * smax;
  const uint32_t rounds = (uint32_t)(((uint64_t)n + per_round - 1u) / per_round);
  const uint32_t groups = slots * rounds;                                  // workgroups
  const uint32_t g = ((uint32_t)n + groups - 1u) / groups;                 // Gaussians per workgroup
  uint32_t planes = (g + 32767u) / 32768u;
  if (planes < 1u) planes = 1u;
  if (planes > smax) planes = smax;
  uint32_t b = ((g + planes - 1u) / planes + 2047u) / 2048u * 2048u;   // (multiples of 2048: a fill workgroup of 1024 threads x 2 stays inside one batch)
  if (b < 4096u) b = 4096u;
  if (b > 32768u) b = 32768u;
  return {b, planes};
}
static uint32_t lr_batches(int32_t n, uint32_t batch) { return batch ? ((uint32_t)n + batch - 1u) / batch : 0u; }
// Does the fill of this view stage the batch's slot-table row in LDS (project.hip: lr_fill_staged_kernel), and with how
// many Gaussians per thread?  0 = no (unbatched, band form, tile grids whose row does not fit: the look-up form).  K x 1024
// consecutive Gaussians of a workgroup share one table row: the largest K <= the knob that divides the batch.  Both stages
// of a forward ask with the same arguments: stage 1 skips lr_rebase_kernel when the fill adds `offsets[]` while staging.
static int lr_fill_staged_k(const LrView& v, uint32_t tiles, uint32_t batch) {
  LR_KNOB(staged_knob, "LOGRAST_FILL_STAGED", 2);
  if (staged_knob <= 0 || batch == 0u || tiles > LR_FILL_STAGED_MAX_TILES || lr_band_sparse(v, (int)batch)) return 0;
  if (batch % LR_FILL_STAGED_ROWS != 0u) return 0;
  const int per_batch = (int)(batch / LR_FILL_STAGED_ROWS);
  int K = staged_knob > 3 ? 3 : staged_knob;
  while (K > 1 && per_batch % K != 0) K--;
  return K;
}
// Two helper passes pay for their launch only on large inputs (each is ~10 us at 1 M Gaussians, where the work they
// save is smaller than that): lr_rebase_kernel (absolute slot table for the fill) and the touched-only clearing of
// dL/dconic.  Both stages of a forward evaluate this with the same n.
static bool lr_big_input(int32_t n) {
  LR_KNOB(min_n, "LOGRAST_HELPER_MIN_N", 4000000);
  return n >= min_n;
}

static int lr_tile_cull() {
  int c = g_tile_cull.load(std::memory_order_relaxed);
  if (c < 0) {
    c = lr_env_int("LOGRAST_TILE_CULL", 1) ? 1 : 0;
    g_tile_cull.store(c, std::memory_order_relaxed);
  }
  return c;
}

// ---- profiling ------------------------------------------------------------------------------------------
static const char* kKernelNames[LOGRAST_NUM_KERNELS] = {
    "compute_radius", "project", "scan_tiles", "fill_keys", "sort_small", "sort_large", "sort_huge",
    "blend_fwd", "blend_bwd", "project_bwd", "knn3", "lod_traverse", "counter_update", "sparse_adam",
    "id_histogram", "gather_activate", "activate_bwd", "count_huge", "rebase_slots", "lazy_tail"};
struct ProfRec { int slot; hipEvent_t a, b; bool own_a; };
// Consecutive launches inside one entry point share an event: the end of kernel k is the begin of kernel k+1 (N+1
// events for a chain of N kernels instead of 2N; every recorded event costs ~1.4 us of stream time).
struct ProfLast { hipStream_t stream; unsigned long long call; hipEvent_t ev; bool valid; };
static thread_local unsigned long long g_prof_call = 0;   // bumped at every entry point that launches kernels
static ProfLast g_prof_last = {nullptr, 0, nullptr, false};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof_open;     // begin recorded, waiting for end
static std::vector<ProfRec> g_prof_done;
static std::vector<hipEvent_t> g_event_pool;
static double g_prof_ms[LOGRAST_NUM_KERNELS];
static int64_t g_prof_cnt[LOGRAST_NUM_KERNELS];
static std::mutex g_prof_mu;

static hipEvent_t lr_get_event() {
  if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
void lr_prof_begin(int slot, hipStream_t s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r{slot, nullptr, lr_get_event(), true};
  if (g_prof_last.valid && g_prof_last.stream == s && g_prof_last.call == g_prof_call) {
    r.a = g_prof_last.ev;   // nothing was enqueued on s since that event: it marks this kernel's begin too
    r.own_a = false;
  } else {
    r.a = lr_get_event();
    (void)hipEventRecord(r.a, s);
  }
  g_prof_last.valid = false;
  g_prof_open.push_back(r);
}
void lr_prof_end(int slot, hipStream_t s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (size_t i = g_prof_open.size(); i-- > 0;) {
    if (g_prof_open[i].slot == slot) {
      (void)hipEventRecord(g_prof_open[i].b, s);
      g_prof_last = ProfLast{s, g_prof_call, g_prof_open[i].b, true};
      g_prof_done.push_back(g_prof_open[i]);
      g_prof_open.erase(g_prof_open.begin() + (long)i);
      return;
    }
  }
}
static void lr_prof_drain_locked() {
  for (auto& r : g_prof_done) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      g_prof_ms[r.slot] += ms;
      g_prof_cnt[r.slot] += 1;
    }
    if (r.own_a) g_event_pool.push_back(r.a);
    g_event_pool.push_back(r.b);
  }
  g_prof_done.clear();
}

static int lr_make_view(const lograst_view* in, LrView* out) {
  if (!in) return lr_fail(LOGRAST_ERR_ARG, "view is NULL");
  if (in->width <= 0 || in->height <= 0) return lr_fail(LOGRAST_ERR_ARG, "image size must be positive");
  if (in->width > 65535 * 16 || in->height > 65535 * 16 ||
      (size_t)((in->width + 15) / 16) * (size_t)((in->height + 15) / 16) > 131072)
    return lr_fail(LOGRAST_ERR_ARG, "image too large (more than 131072 tiles)");
  if (!in->viewmatrix || !in->projmatrix || !in->bg) return lr_fail(LOGRAST_ERR_ARG, "viewmatrix/projmatrix/bg must be device pointers");
  if (in->filter_mode < 0 || in->filter_mode > 2) return lr_fail(LOGRAST_ERR_ARG, "bad filter_mode");
  out->W = in->width; out->H = in->height;
  out->gx = (in->width + LOGRAST_TILE - 1) / LOGRAST_TILE;
  out->gy = (in->height + LOGRAST_TILE - 1) / LOGRAST_TILE;
  out->ty0 = 0; out->ty1 = out->gy;
  if (in->tile_row_begin != 0 || in->tile_row_end != 0) {
    if (in->tile_row_begin < 0 || in->tile_row_end <= in->tile_row_begin || in->tile_row_end > out->gy)
      return lr_fail(LOGRAST_ERR_ARG, "tile_row_begin / tile_row_end outside the tile grid");
    out->ty0 = in->tile_row_begin; out->ty1 = in->tile_row_end;
  }
  out->tanfovx = in->tanfovx; out->tanfovy = in->tanfovy;
  out->fx = (float)in->width / (2.0f * in->tanfovx);
  out->fy = (float)in->height / (2.0f * in->tanfovy);
  out->scale_modifier = in->scale_modifier;
  out->filter_mode = in->filter_mode; out->ndc_cull = in->ndc_cull; out->extras = in->extras;
  out->view = in->viewmatrix; out->proj = in->projmatrix; out->bg = in->bg;
  out->cov3d = in->cov3d_precomp; out->g_cov3d = in->dl_dcov3d;
  if (in->walk_form < LOGRAST_FORM_AUTO || in->walk_form > LOGRAST_FORM_QUADRANT) return lr_fail(LOGRAST_ERR_ARG, "bad walk_form");
  out->walk_form = in->walk_form;
  LR_KNOB(hit_masks, "LOGRAST_HIT_MASKS", 1);
  if (in->hit_masks && (reinterpret_cast<uintptr_t>(in->hit_masks) & 31u)) return lr_fail(LOGRAST_ERR_ARG, "hit_masks must be 32-byte aligned");
  out->masks = hit_masks ? in->hit_masks : nullptr;
  out->mask_words = in->hit_mask_words;
  if (in->hit_mask_form < 0 || in->hit_mask_form > 2) return lr_fail(LOGRAST_ERR_ARG, "bad hit_mask_form");
  out->mask_form = in->hit_mask_form;
  return LOGRAST_OK;
}

extern "C" {

int lograst_version(void) { return LOGRAST_VERSION; }
const char* lograst_last_error(void) { return g_err.c_str(); }

size_t lograst_tile_state_bytes(int32_t width, int32_t height, int32_t n) {
  uint32_t gx = (uint32_t)(width + LOGRAST_TILE - 1) / LOGRAST_TILE, gy = (uint32_t)(height + LOGRAST_TILE - 1) / LOGRAST_TILE;
  return sizeof(uint32_t) * lr_state_words(gx * gy, lr_batches(n, lr_pick_batch(n, gx * gy, gx, gy).batch));
}
size_t lograst_geom_bytes(int32_t n) {  // 64-byte records + the 16-byte fill records of the batched projection + a 4-byte index each (band views) + the rank rows of the 5..16-tile rects (common.hpp)
  const size_t nn = (size_t)(n > 0 ? n : 0);
  return lr_midrank_off_bytes(nn) + lr_midrank_bytes(nn);
}
size_t lograst_keys_bytes(uint32_t capacity) { return 2 * sizeof(uint64_t) * (size_t)capacity; }  // keys + sort scratch
int lograst_forward_form(const lograst_view* view) {
  LrView v;
  int rc = lr_make_view(view, &v);
  if (rc) return rc;
  return lr_blend_fwd_form(v);
}
size_t lograst_hit_mask_bytes(uint32_t capacity, int32_t width, int32_t height) {   // blend.hip: 16 words per (tile, 64-entry chunk) slot
  const size_t gx = (size_t)(width > 0 ? (width + LOGRAST_TILE - 1) / LOGRAST_TILE : 0), gy = (size_t)(height > 0 ? (height + LOGRAST_TILE - 1) / LOGRAST_TILE : 0);
  return 16 * sizeof(uint64_t) * ((size_t)capacity / 64 + gx * gy + 1);
}
size_t lograst_list_bytes(uint32_t capacity) { return sizeof(uint32_t) * (size_t)capacity; }

const uint32_t* lograst_tile_offsets(const void* tile_state, int32_t width, int32_t height) {
  uint32_t gx = (uint32_t)(width + LOGRAST_TILE - 1) / LOGRAST_TILE, gy = (uint32_t)(height + LOGRAST_TILE - 1) / LOGRAST_TILE;
  return reinterpret_cast<const uint32_t*>(tile_state) + lr_offsets_off(gx * gy);
}

// Lazily ordered lists (LOGRAST_LAZY_SORT; common.hpp: sorted[]): how much of every tile's list is in final order, and
// the call that orders the rest -- for callers that want the complete lists (the parity tests do).
int lograst_ordered_lengths(const void* tile_state, int32_t width, int32_t height, uint32_t* lengths_out, void* stream) {
  g_prof_call++;
  if (!tile_state || !lengths_out) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  if (width <= 0 || height <= 0) return lr_fail(LOGRAST_ERR_ARG, "bad image size");
  const uint32_t gx = (uint32_t)(width + LOGRAST_TILE - 1) / LOGRAST_TILE, gy = (uint32_t)(height + LOGRAST_TILE - 1) / LOGRAST_TILE;
  lr_launch_ordered_lengths(reinterpret_cast<const uint32_t*>(tile_state), gx * gy, lengths_out, (hipStream_t)stream);
  LR_HIP(hipGetLastError());
  return LOGRAST_OK;
}
int lograst_finish_lists(void* tile_state, int32_t width, int32_t height, void* keys, uint32_t* point_list,
                         uint32_t capacity, void* stream) {
  g_prof_call++;
  if (!tile_state) return lr_fail(LOGRAST_ERR_ARG, "tile_state is NULL");
  if (width <= 0 || height <= 0) return lr_fail(LOGRAST_ERR_ARG, "bad image size");
  if (capacity == 0) return LOGRAST_OK;
  if (!keys || !point_list) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  const uint32_t gx = (uint32_t)(width + LOGRAST_TILE - 1) / LOGRAST_TILE, gy = (uint32_t)(height + LOGRAST_TILE - 1) / LOGRAST_TILE;
  // The forward documents `keys` as dead after the call; what this entry point orders from must be the very buffer that
  // forward's fill wrote (round-5 advisory: a stale or reused buffer silently corrupted point_list and marked it ordered).
  // The fill left (pointer, capacity) in the header; this diagnostic call reads them back (it synchronises `stream`).
  uint32_t hdr[LR_HDR_WORDS];
  LR_HIP(hipMemcpyAsync(hdr, tile_state, sizeof(hdr), hipMemcpyDeviceToHost, (hipStream_t)stream));
  LR_HIP(hipStreamSynchronize((hipStream_t)stream));
  const uint64_t kp = (uint64_t)reinterpret_cast<uintptr_t>(keys);
  if (hdr[LR_HDR_KEYS_LO] != (uint32_t)kp || hdr[LR_HDR_KEYS_HI] != (uint32_t)(kp >> 32) || hdr[LR_HDR_KEYS_CAP] != capacity)
    return lr_fail(LOGRAST_ERR_ARG, "keys / capacity are not the buffer this tile_state's forward filled");
  lr_launch_sort_rest(reinterpret_cast<uint32_t*>(tile_state), gx * gy, reinterpret_cast<uint64_t*>(keys), point_list,
                      capacity, 0, 3, (hipStream_t)stream);
  LR_HIP(hipGetLastError());
  return LOGRAST_OK;
}

int lograst_compute_radius(int32_t p, const float* means3d, const float* scales, const float* rotations,
                           const float* projmatrix, const float* viewmatrix, float focal_x, float focal_y,
                           float tanfovx, float tanfovy, float* radii_out, void* stream) {
  g_prof_call++;
  if (p < 0) return lr_fail(LOGRAST_ERR_ARG, "negative point count");
  if (p == 0) return LOGRAST_OK;
  if (!means3d || !scales || !rotations || !projmatrix || !viewmatrix || !radii_out)
    return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  lr_launch_radius(p, means3d, scales, rotations, projmatrix, viewmatrix, focal_x, focal_y, tanfovx, tanfovy,
                   radii_out, (hipStream_t)stream);
  LR_HIP(hipGetLastError());
  return LOGRAST_OK;
}

// stage 1 launches: memset of header + counters, projection (+ counting / ranking), tile scan
static int lr_stage1(const LrView& v, int32_t n, const float* means3d, const float* scales, const float* rotations,
                     const float* opacities, const float* colors, int32_t* radii, void* geom, uint32_t* st,
                     hipStream_t s) {
  const uint32_t tiles = (uint32_t)(v.gx * v.gy);
  const LrBatching bt = lr_pick_batch(n, tiles, (uint32_t)v.gx, (uint32_t)v.gy);
  // Counters: batched projection -> dense (ranked[tiles] | big[tiles] right behind the header), unbatched -> one
  // counter per 64 B.  Header and counters are zeroed by ONE memset (offsets/cursors are fully rewritten by the scan).
  const uint32_t cs = bt.batch ? 1u : (uint32_t)LR_CTR_STRIDE;
  const uint32_t big_off = bt.batch ? lr_ranked_off(tiles) + tiles : lr_big_off(tiles);
  lr_launch_zero_words(st, ((size_t)(big_off + tiles * cs) + 3) & ~(size_t)3, s);   // the words behind the counters (offsets[]) are rewritten by the scan
  lr_launch_project(v, n, means3d, scales, rotations, opacities, colors, radii, geom, st + lr_ranked_off(tiles),
                    st + big_off, st, st + lr_basetab_off(tiles), (int)bt.batch, (int)bt.planes, lr_tile_cull(), s);
  lr_launch_scan(st, tiles, cs, big_off, s);
  if (lr_big_input(n) && lr_fill_staged_k(v, tiles, bt.batch) == 0) {   // (the staged fill adds offsets[] itself)
    const bool band = lr_band_sparse(v, (int)bt.batch);   // only the band's tiles have slot-table entries
    lr_launch_rebase(st, tiles, lr_batches(n, bt.batch), band ? (uint32_t)(v.ty0 * v.gx) : 0u,
                     band ? (uint32_t)(v.ty1 * v.gx) : tiles, s);
  }
  return LOGRAST_OK;
}

// (a list is streamed -- and may be left at its first window -- only above LR_LONG_LIST keys: with a smaller bound on the
// longest list known to the host the lazy machinery is not launched at all)
static inline bool max_tile_len_allows_streaming(uint32_t max_tile_len, uint32_t capacity) {
  const uint32_t m = (max_tile_len == 0 || max_tile_len > capacity) ? capacity : max_tile_len;
  return m > LR_LONG_LIST;
}

// stage 2 launches: bucket fill (+ zero-fills), per-tile sort, compositing
static int lr_stage2(const LrView& v, int32_t n, const void* geom, uint32_t* st, uint64_t* keys, uint32_t* point_list,
                     uint32_t capacity, uint32_t max_tile_len, float* image, float* final_t, int32_t* n_contrib,
                     int32_t* point_id_pixel, float* point_weight_pixel, float* point_weight, float* bwd_scratch,
                     int32_t bwd_scratch_floats, uint32_t* status, hipStream_t s, int speculative = 0) {
  const uint32_t tiles = (uint32_t)(v.gx * v.gy);
  static const int stop_after_project = LR_EXPERIMENT_INT("LOGRAST_STOP_AFTER_PROJECT", 0);   // experiment builds (tools/kernel_probe.py)
  if (stop_after_project) return LOGRAST_OK;
  if (n == 0 && status)   // no fill kernel runs: this forward's entries of the status block
    LR_HIP(hipMemsetAsync(status + LOGRAST_STATUS_LAST_INSTANCES, 0, 4 * sizeof(uint32_t), s));
  // point_weight (atomicMax target) and the optional backward scratch (one 64-byte accumulator row per Gaussian) are
  // cleared by the fill kernel -- except, in the 5-tuple flavour on large inputs, the scratch: only the rows of
  // Gaussians that contribute to a pixel will ever be read, and the compositing kernel clears exactly those when it
  // meets them (a separate pass over point_weight afterwards cost 70 us per 30 M-Gaussian view, the stores inside the
  // kernel 30)
  const bool touched_only = v.extras && bwd_scratch_floats > 0 && lr_big_input(n);
  float* zero_block = bwd_scratch_floats > 0 ? bwd_scratch : nullptr;
  int zero_floats = bwd_scratch_floats;
  if (touched_only) { zero_block = nullptr; zero_floats = 0; }
  float* zero_n = v.extras ? point_weight : nullptr;
  LR_KNOB(separate_zero, "LOGRAST_SEPARATE_ZERO", 1);   // 0: always inside the fill kernel
  if (separate_zero && lr_big_input(n)) {   // large inputs: streamed by kernels of their own (see lr_zero_floats_kernel)
    lr_launch_zero_floats(zero_n, (size_t)n, s);
    if (zero_floats > 0) lr_launch_zero_floats(zero_block, (size_t)zero_floats * (size_t)n, s);
    zero_n = nullptr; zero_floats = 0;
  }
  const uint32_t fill_batch = lr_pick_batch(n, tiles, (uint32_t)v.gx, (uint32_t)v.gy).batch;
  const int staged_k = lr_fill_staged_k(v, tiles, fill_batch);
  lr_launch_fill(n, v.gx, geom, st, tiles, keys, capacity, max_tile_len, status,
                 zero_n, zero_floats > 0 ? zero_block : nullptr, zero_floats,
                 (lr_big_input(n) && staged_k == 0) ? 1 : 0, speculative, lr_band_sparse(v, (int)fill_batch) ? 1 : 0, staged_k, s);
  static const int stop_after_fill = LR_EXPERIMENT_INT("LOGRAST_STOP_AFTER_FILL", 0);   // experiment builds (tools/fill_probe.py)
  if (stop_after_fill) return LOGRAST_OK;
  // LOGRAST_LAZY_SORT (default 1): lists of more than 4096 keys are ordered over their first window only (7680 positions;
  // the walk of a view ends far in front of that: common.hpp, sorted[]); the compositing kernels mark the tiles that needed
  // more, and the second pair of launches -- idle in every benched view -- finishes exactly those.  0: every list to its
  // end before the first compositing pass (what lograst_finish_lists produces afterwards).
  LR_KNOB(lazy_knob, "LOGRAST_LAZY_SORT", 1);
  const int lazy_asked = (lazy_knob && max_tile_len_allows_streaming(max_tile_len, capacity)) ? 1 : 0;
  const int lazy = lr_launch_sort(st, tiles, keys, point_list, capacity, max_tile_len, lazy_asked, s);
  float* const zrows = touched_only ? bwd_scratch : nullptr;
  lr_launch_blend_fwd(v, geom, st, tiles, point_list, capacity, image, final_t, n_contrib, point_id_pixel,
                      point_weight_pixel, point_weight, zrows, lr_big_input(n) ? 1 : 0, lazy, v.masks, s);
  if (lazy) {
    lr_prof_begin(LRK_LAZY_TAIL, s);
    lr_launch_sort_rest(st, tiles, keys, point_list, capacity, max_tile_len, 2, s);
    lr_launch_blend_fwd(v, geom, st, tiles, point_list, capacity, image, final_t, n_contrib, point_id_pixel,
                        point_weight_pixel, point_weight, zrows, lr_big_input(n) ? 1 : 0, 2, v.masks, s);
    lr_prof_end(LRK_LAZY_TAIL, s);
  }
  return LOGRAST_OK;
}

static int lr_check_stage1_args(const LrView& v, int32_t n, const float* means3d, const float* scales,
                                const float* rotations, const float* opacities, const float* colors,
                                const int32_t* radii, const void* geom, const void* tile_state) {
  if (n < 0) return lr_fail(LOGRAST_ERR_ARG, "negative Gaussian count");
  if (!tile_state) return lr_fail(LOGRAST_ERR_ARG, "tile_state is NULL");
  if (n > 0 && (!means3d || !opacities || !colors || !radii || !geom))
    return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  if (n > 0 && !v.cov3d && (!scales || !rotations))
    return lr_fail(LOGRAST_ERR_ARG, "scales / rotations are NULL and the view carries no cov3d_precomp");
  if ((reinterpret_cast<uintptr_t>(rotations) | reinterpret_cast<uintptr_t>(geom) | reinterpret_cast<uintptr_t>(tile_state)) & 15u)
    return lr_fail(LOGRAST_ERR_ARG, "rotations / geom / tile_state must be 16-byte aligned");
  return LOGRAST_OK;
}

// backward: scales + rotations with their gradient outputs, or the view's cov3d_precomp with dl_dcov3d
static int lr_check_cov_args(const LrView& v, const float* scales, const float* rotations, const float* dl_dscales,
                             const float* dl_drotations) {
  if (v.cov3d) {
    if (!v.g_cov3d) return lr_fail(LOGRAST_ERR_ARG, "cov3d_precomp is set but dl_dcov3d is NULL");
    return LOGRAST_OK;
  }
  if (!scales || !rotations || !dl_dscales || !dl_drotations) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  return LOGRAST_OK;
}

static int lr_check_stage2_args(const LrView& v, int32_t n, const void* tile_state, const uint64_t* keys,
                                const uint32_t* point_list, uint32_t capacity, const float* image, const float* final_t,
                                const int32_t* n_contrib, const int32_t* point_id_pixel, const float* point_weight_pixel,
                                const float* point_weight, const float* bwd_scratch, int32_t bwd_scratch_floats) {
  if (n < 0) return lr_fail(LOGRAST_ERR_ARG, "negative Gaussian count");
  if (!tile_state || !image || !final_t || !n_contrib) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  if (capacity > 0 && (!keys || !point_list)) return lr_fail(LOGRAST_ERR_ARG, "keys/point_list NULL with capacity > 0");
  if (v.extras && (!point_id_pixel || !point_weight_pixel || (n > 0 && !point_weight)))
    return lr_fail(LOGRAST_ERR_ARG, "extras requested but output pointers are NULL");
  if ((bwd_scratch_floats != 0 && bwd_scratch_floats != LOGRAST_BWD_ROW_FLOATS) ||
      (bwd_scratch_floats > 0 && n > 0 && !bwd_scratch))
    return lr_fail(LOGRAST_ERR_ARG, "bwd_scratch: 0 or LOGRAST_BWD_ROW_FLOATS (16) floats per Gaussian and a non-NULL block");
  if (bwd_scratch_floats > 0 && (reinterpret_cast<uintptr_t>(bwd_scratch) & 63u))
    return lr_fail(LOGRAST_ERR_ARG, "bwd_scratch must be 64-byte aligned (one accumulator row per line)");
  if (v.masks && (size_t)v.mask_words * sizeof(uint64_t) < lograst_hit_mask_bytes(capacity, v.W, v.H))
    return lr_fail(LOGRAST_ERR_ARG, "hit_mask_words is smaller than lograst_hit_mask_bytes(capacity, width, height) / 8");
  return LOGRAST_OK;
}

int lograst_forward_project(const lograst_view* view, int32_t n, const float* means3d, const float* scales,
                            const float* rotations, const float* opacities, const float* colors,
                            int32_t* radii, void* geom, void* tile_state, uint32_t* num_instances_host,
                            uint32_t* max_tile_len_host, void* stream) {
  g_prof_call++;
  LrView v;
  int rc = lr_make_view(view, &v);
  if (rc) return rc;
  rc = lr_check_stage1_args(v, n, means3d, scales, rotations, opacities, colors, radii, geom, tile_state);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  uint32_t* st = reinterpret_cast<uint32_t*>(tile_state);
  rc = lr_stage1(v, n, means3d, scales, rotations, opacities, colors, radii, geom, st, s);
  if (rc) return rc;
  LR_HIP(hipGetLastError());
  if (num_instances_host || max_tile_len_host) {
    uint32_t hdr[LR_HDR_WORDS] = {0};
    LR_HIP(hipMemcpyAsync(hdr, st, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, s));
    LR_HIP(hipStreamSynchronize(s));
    if (num_instances_host) *num_instances_host = hdr[LR_HDR_NUM];
    if (max_tile_len_host) *max_tile_len_host = hdr[LR_HDR_MAXLEN];
  }
  return LOGRAST_OK;
}

int lograst_forward_render(const lograst_view* view, int32_t n, const void* geom, void* tile_state,
                           uint64_t* keys, uint32_t* point_list, uint32_t capacity, uint32_t max_tile_len,
                           float* image, float* final_t, int32_t* n_contrib, int32_t* point_id_pixel,
                           float* point_weight_pixel, float* point_weight, float* bwd_scratch,
                           int32_t bwd_scratch_floats, uint32_t* status, void* stream) {
  g_prof_call++;
  LrView v;
  int rc = lr_make_view(view, &v);
  if (rc) return rc;
  rc = lr_check_stage2_args(v, n, tile_state, keys, point_list, capacity, image, final_t, n_contrib, point_id_pixel,
                            point_weight_pixel, point_weight, bwd_scratch, bwd_scratch_floats);
  if (rc) return rc;
  rc = lr_stage2(v, n, geom, reinterpret_cast<uint32_t*>(tile_state), keys, point_list, capacity, max_tile_len, image,
                 final_t, n_contrib, point_id_pixel, point_weight_pixel, point_weight, bwd_scratch, bwd_scratch_floats,
                 status, (hipStream_t)stream);
  if (rc) return rc;
  LR_HIP(hipGetLastError());
  return LOGRAST_OK;
}

int lograst_forward(const lograst_view* view, int32_t n, const float* means3d, const float* scales,
                    const float* rotations, const float* opacities, const float* colors, int32_t* radii, void* geom,
                    void* tile_state, uint64_t* keys, uint32_t* point_list, uint32_t capacity, uint32_t max_tile_len,
                    float* image, float* final_t, int32_t* n_contrib, int32_t* point_id_pixel,
                    float* point_weight_pixel, float* point_weight, float* bwd_scratch, int32_t bwd_scratch_floats,
                    uint32_t* status, void* stream) {
  g_prof_call++;
  LrView v;
  int rc = lr_make_view(view, &v);
  if (rc) return rc;
  rc = lr_check_stage1_args(v, n, means3d, scales, rotations, opacities, colors, radii, geom, tile_state);
  if (rc) return rc;
  rc = lr_check_stage2_args(v, n, tile_state, keys, point_list, capacity, image, final_t, n_contrib, point_id_pixel,
                            point_weight_pixel, point_weight, bwd_scratch, bwd_scratch_floats);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  uint32_t* st = reinterpret_cast<uint32_t*>(tile_state);
  rc = lr_stage1(v, n, means3d, scales, rotations, opacities, colors, radii, geom, st, s);
  if (rc) return rc;
  rc = lr_stage2(v, n, geom, st, keys, point_list, capacity, max_tile_len, image, final_t, n_contrib, point_id_pixel,
                 point_weight_pixel, point_weight, bwd_scratch, bwd_scratch_floats, status, s);
  if (rc) return rc;
  LR_HIP(hipGetLastError());
  return LOGRAST_OK;
}

// Side stream + event + pinned words for the read-back of lograst_forward_speculative, one set per host thread and
// device (created on first use, never destroyed: they live as long as the process).
struct LrSpec { int dev; hipStream_t side; hipEvent_t ev; uint32_t* pinned; };
static thread_local std::vector<LrSpec> g_spec;
static int lr_spec_get(LrSpec** out) {
  int dev = 0;
  LR_HIP(hipGetDevice(&dev));
  for (auto& e : g_spec)
    if (e.dev == dev) { *out = &e; return LOGRAST_OK; }
  LrSpec e{dev, nullptr, nullptr, nullptr};
  LR_HIP(hipStreamCreateWithFlags(&e.side, hipStreamNonBlocking));
  LR_HIP(hipEventCreateWithFlags(&e.ev, hipEventDisableTiming));
  LR_HIP(hipHostMalloc(reinterpret_cast<void**>(&e.pinned), sizeof(uint32_t) * LR_HDR_WORDS, hipHostMallocDefault));
  g_spec.push_back(e);
  *out = &g_spec.back();
  return LOGRAST_OK;
}

int lograst_forward_speculative(const lograst_view* view, int32_t n, const float* means3d, const float* scales,
                                const float* rotations, const float* opacities, const float* colors, int32_t* radii,
                                void* geom, void* tile_state, uint64_t* keys, uint32_t* point_list, uint32_t capacity,
                                uint32_t max_tile_len, float* image, float* final_t, int32_t* n_contrib,
                                int32_t* point_id_pixel, float* point_weight_pixel, float* point_weight,
                                float* bwd_scratch, int32_t bwd_scratch_floats, uint32_t* status,
                                uint32_t* num_instances_host, uint32_t* max_tile_len_host, void* stream) {
  g_prof_call++;
  LrView v;
  int rc = lr_make_view(view, &v);
  if (rc) return rc;
  rc = lr_check_stage1_args(v, n, means3d, scales, rotations, opacities, colors, radii, geom, tile_state);
  if (rc) return rc;
  rc = lr_check_stage2_args(v, n, tile_state, keys, point_list, capacity, image, final_t, n_contrib, point_id_pixel,
                            point_weight_pixel, point_weight, bwd_scratch, bwd_scratch_floats);
  if (rc) return rc;
  if (!num_instances_host || !max_tile_len_host) return lr_fail(LOGRAST_ERR_ARG, "NULL host pointer");
  hipStream_t s = (hipStream_t)stream;
  uint32_t* st = reinterpret_cast<uint32_t*>(tile_state);
  LrSpec* sp = nullptr;
  rc = lr_spec_get(&sp);
  if (rc) return rc;
  rc = lr_stage1(v, n, means3d, scales, rotations, opacities, colors, radii, geom, st, s);
  if (rc) return rc;
  LR_HIP(hipEventRecord(sp->ev, s));   // the scan has written the header: instance count and longest list
  // stage 2 is enqueued before the host knows whether `capacity` suffices: the stream never waits for the host
  rc = lr_stage2(v, n, geom, st, keys, point_list, capacity, max_tile_len, image, final_t, n_contrib, point_id_pixel,
                 point_weight_pixel, point_weight, bwd_scratch, bwd_scratch_floats, status, s, 1);
  if (rc) return rc;
  LR_HIP(hipGetLastError());
  // ... and the header is read on a side stream that waits for stage 1 only (the fill kernel rewrites only the overflow
  // word, which is not read here)
  LR_HIP(hipStreamWaitEvent(sp->side, sp->ev, 0));
  LR_HIP(hipMemcpyAsync(sp->pinned, st, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, sp->side));
  LR_HIP(hipStreamSynchronize(sp->side));
  *num_instances_host = sp->pinned[LR_HDR_NUM];
  *max_tile_len_host = sp->pinned[LR_HDR_MAXLEN];
  return LOGRAST_OK;
}

int lograst_tile_rows(const lograst_view* view, int32_t n, const float* means3d, const float* scales,
                      const float* rotations, uint32_t* rows_out, void* stream) {
  g_prof_call++;
  LrView v;
  int rc = lr_make_view(view, &v);
  if (rc) return rc;
  if (n < 0) return lr_fail(LOGRAST_ERR_ARG, "negative Gaussian count");
  if (n == 0) return LOGRAST_OK;
  if (!means3d || !rows_out || (!v.cov3d && (!scales || !rotations))) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  if (reinterpret_cast<uintptr_t>(rotations) & 15u) return lr_fail(LOGRAST_ERR_ARG, "rotations must be 16-byte aligned");
  v.ty0 = 0; v.ty1 = v.gy;   // always the rows of the whole image: the caller intersects them with its band
  lr_launch_tile_rows(v, n, means3d, scales, rotations, rows_out, (hipStream_t)stream);
  LR_HIP(hipGetLastError());
  return LOGRAST_OK;
}

size_t lograst_sparse_segment_floats(int32_t kmax) {
  const size_t k = kmax > 0 ? (size_t)kmax : 0;
  return 16 + 16 * k + ((k + 15) / 16) * 16;
}
static int lr_pack_rows_checked(const char* who, float* rows, int32_t groups, int64_t rows_per_group, int32_t kmax, float* packed,
                                uint32_t* overflow, int clear, const uint32_t* hint, int64_t hint_rows, void* stream) {
  if (groups < 0 || rows_per_group < 0 || kmax <= 0 || hint_rows < 0) return lr_fail(LOGRAST_ERR_ARG, std::string(who) + ": negative size or kmax <= 0");
  if (groups == 0 || rows_per_group == 0) return LOGRAST_OK;
  if (!rows || !packed) return lr_fail(LOGRAST_ERR_ARG, std::string(who) + ": NULL pointer");
  if ((reinterpret_cast<uintptr_t>(rows) | reinterpret_cast<uintptr_t>(packed)) & 15u)
    return lr_fail(LOGRAST_ERR_ARG, std::string(who) + ": rows / packed must be 16-byte aligned");
  if (rows_per_group > 0x7fffffffLL) return lr_fail(LOGRAST_ERR_ARG, std::string(who) + ": rows_per_group exceeds 31 bits (int32 row indices)");
  lx_launch_pack_rows(rows, groups, rows_per_group, kmax, packed, lograst_sparse_segment_floats(kmax), overflow, clear, hint,
                      hint_rows, (hipStream_t)stream);
  LR_HIP(hipGetLastError());
  return LOGRAST_OK;
}
int lograst_pack_rows(const float* rows, int32_t groups, int64_t rows_per_group, int32_t kmax, float* packed,
                      uint32_t* overflow, void* stream) {
  return lr_pack_rows_checked("lograst_pack_rows", const_cast<float*>(rows), groups, rows_per_group, kmax, packed, overflow, 0, nullptr, 0, stream);
}
int lograst_pack_rows_clear(float* rows, int32_t groups, int64_t rows_per_group, int32_t kmax, float* packed,
                            uint32_t* overflow, void* stream) {
  return lr_pack_rows_checked("lograst_pack_rows_clear", rows, groups, rows_per_group, kmax, packed, overflow, 1, nullptr, 0, stream);
}
int lograst_pack_rows_hinted(float* rows, int32_t groups, int64_t rows_per_group, int32_t kmax, float* packed,
                             uint32_t* overflow, int32_t clear, const uint32_t* hint, int64_t hint_rows, void* stream) {
  if (!hint) return lr_fail(LOGRAST_ERR_ARG, "lograst_pack_rows_hinted: NULL hint (use lograst_pack_rows / lograst_pack_rows_clear)");
  return lr_pack_rows_checked("lograst_pack_rows_hinted", rows, groups, rows_per_group, kmax, packed, overflow, clear ? 1 : 0, hint,
                              hint_rows, stream);
}
int lograst_add_visible(float* seen, const int32_t* radii, int64_t n, void* stream) {
  if (n < 0) return lr_fail(LOGRAST_ERR_ARG, "lograst_add_visible: negative n");
  if (n == 0) return LOGRAST_OK;
  if (!seen || !radii) return lr_fail(LOGRAST_ERR_ARG, "lograst_add_visible: NULL pointer");
  lx_launch_add_visible(seen, radii, n, (hipStream_t)stream);
  LR_HIP(hipGetLastError());
  return LOGRAST_OK;
}
int lograst_add_visible_n(float* seen, const int32_t* const* radii, int32_t k, int64_t n, void* stream) {
  if (n < 0 || k < 0 || k > 16) return lr_fail(LOGRAST_ERR_ARG, "lograst_add_visible_n: negative n, or k outside 0..16");
  if (n == 0 || k == 0) return LOGRAST_OK;
  if (!seen || !radii) return lr_fail(LOGRAST_ERR_ARG, "lograst_add_visible_n: NULL pointer");
  for (int j = 0; j < k; j++)
    if (!radii[j]) return lr_fail(LOGRAST_ERR_ARG, "lograst_add_visible_n: NULL radii pointer");
  lx_launch_add_visible_n(seen, radii, k, n, (hipStream_t)stream);
  LR_HIP(hipGetLastError());
  return LOGRAST_OK;
}
int lograst_unpack_rows(float* dest, const float* packed, int32_t segments, int32_t kmax, int64_t rows_per_group,
                        int64_t dest_group_rows, int32_t atomic, void* stream) {
  if (segments < 0 || kmax <= 0 || rows_per_group < 0 || dest_group_rows < 0)
    return lr_fail(LOGRAST_ERR_ARG, "lograst_unpack_rows: negative size or kmax <= 0");
  if (segments == 0 || rows_per_group == 0) return LOGRAST_OK;
  if (!dest || !packed) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  if ((reinterpret_cast<uintptr_t>(dest) | reinterpret_cast<uintptr_t>(packed)) & 15u)
    return lr_fail(LOGRAST_ERR_ARG, "lograst_unpack_rows: dest / packed must be 16-byte aligned");
  if (atomic != 1 && dest_group_rows < rows_per_group)
    return lr_fail(LOGRAST_ERR_ARG, "lograst_unpack_rows: dest_group_rows must cover rows_per_group (segment s owns rows [s * dest_group_rows, ...))");
  if (atomic < 0 || atomic > 2) return lr_fail(LOGRAST_ERR_ARG, "lograst_unpack_rows: atomic is 0 (store), 1 (add) or 2 (zero the named rows)");
  lx_launch_unpack_rows(dest, packed, segments, kmax, lograst_sparse_segment_floats(kmax), rows_per_group,
                        atomic == 1 ? 0 : dest_group_rows, atomic == 1, atomic == 2, (hipStream_t)stream);
  LR_HIP(hipGetLastError());
  return LOGRAST_OK;
}

int lograst_stream_copy(void* dst, const void* src, size_t bytes, int32_t blocks, void* stream) {
  if (bytes && (!dst || !src)) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | bytes) & 15u)
    return lr_fail(LOGRAST_ERR_ARG, "lograst_stream_copy: pointers and size must be multiples of 16 bytes");
  lr_launch_stream_copy(src, dst, bytes, blocks, (hipStream_t)stream);
  LR_HIP(hipGetLastError());
  return LOGRAST_OK;
}

int lograst_knob_count(void) { return kNumKnobs; }
int lograst_knob_info(int32_t index, const char** name, int32_t* dflt, int32_t* lo, int32_t* hi, const char** what) {
  if (index < 0 || index >= kNumKnobs) return lr_fail(LOGRAST_ERR_ARG, "knob index out of range");
  if (name) *name = kKnobs[index].name;
  if (dflt) *dflt = kKnobs[index].dflt;
  if (lo) *lo = kKnobs[index].lo;
  if (hi) *hi = kKnobs[index].hi;
  if (what) *what = kKnobs[index].what;
  return LOGRAST_OK;
}
static const LrKnobInfo* lr_find_knob(const char* name) {
  if (!name) return nullptr;
  for (int i = 0; i < kNumKnobs; i++)
    if (std::strcmp(kKnobs[i].name, name) == 0) return &kKnobs[i];
  return nullptr;
}
int lograst_set_knob(const char* name, int32_t value) {
  const LrKnobInfo* k = lr_find_knob(name);
  if (!k) return lr_fail(LOGRAST_ERR_ARG, std::string("unknown knob: ") + (name ? name : "(null)"));
  if (value < k->lo || value > k->hi) return lr_fail(LOGRAST_ERR_ARG, std::string(name) + ": value out of range");
  {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    bool found = false;
    for (auto& kv : g_knob_over)
      if (kv.first == name) { kv.second = value; found = true; }
    if (!found) g_knob_over.emplace_back(name, value);
  }
  g_knob_gen.fetch_add(1, std::memory_order_acq_rel);
  return LOGRAST_OK;
}
int lograst_get_knob(const char* name, int32_t* value) {
  const LrKnobInfo* k = lr_find_knob(name);
  if (!k || !value) return lr_fail(LOGRAST_ERR_ARG, "unknown knob or NULL pointer");
  *value = lr_knob_lookup(k->name, k->dflt);
  return LOGRAST_OK;
}
int lograst_reset_knobs(void) {
  {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    g_knob_over.clear();
  }
  g_knob_gen.fetch_add(1, std::memory_order_acq_rel);
  return LOGRAST_OK;
}

int lograst_set_tile_cull(int enabled) {
  const int old = lr_tile_cull();
  g_tile_cull.store(enabled ? 1 : 0, std::memory_order_relaxed);
  return old;
}

int lograst_read_state(const void* tile_state, uint32_t* num_instances_host, uint32_t* overflow_host,
                       uint32_t* max_tile_len_host, uint32_t* rect_instances_host, void* stream) {
  if (!tile_state) return lr_fail(LOGRAST_ERR_ARG, "tile_state is NULL");
  uint32_t hdr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hipStream_t s = (hipStream_t)stream;
  LR_HIP(hipMemcpyAsync(hdr, tile_state, sizeof(hdr), hipMemcpyDeviceToHost, s));
  LR_HIP(hipStreamSynchronize(s));
  if (num_instances_host) *num_instances_host = hdr[LR_HDR_NUM];
  if (overflow_host) *overflow_host = hdr[LR_HDR_OVERFLOW];
  if (max_tile_len_host) *max_tile_len_host = hdr[LR_HDR_MAXLEN];
  if (rect_instances_host) *rect_instances_host = hdr[LR_HDR_RECT];
  return LOGRAST_OK;
}

int lograst_backward(const lograst_view* view, int32_t n, const float* means3d, const float* scales,
                     const float* rotations, const int32_t* radii, const void* geom, const void* tile_state,
                     const uint32_t* point_list, const float* final_t, const int32_t* n_contrib,
                     const float* dl_dimage, float* dl_dmeans2d, float* bwd_rows, float* dl_dopacities,
                     float* dl_dcolors, float* dl_dmeans3d, float* dl_dscales, float* dl_drotations,
                     const float* point_weight, int32_t flags, void* stream) {
  float* const dl_dconic = bwd_rows;   // (the fourth gradient argument: since version 3 the n x 16 accumulator rows)
  g_prof_call++;
  LrView v;
  int rc = lr_make_view(view, &v);
  if (rc) return rc;
  if (n < 0) return lr_fail(LOGRAST_ERR_ARG, "negative Gaussian count");
  if (n == 0) return LOGRAST_OK;
  const bool sink_rows = (flags & LOGRAST_BWD_ACCUMULATE_ROWS) != 0;
  if (!means3d || !radii || !geom || !tile_state || !final_t || !n_contrib || !dl_dimage ||
      !dl_dmeans2d || !dl_dconic || !dl_dmeans3d || (!sink_rows && (!dl_dopacities || !dl_dcolors)))
    return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  if (sink_rows) {   // dl_dmeans3d = the caller's [n][LOGRAST_GRAD_ROW_FLOATS] running sums; the other four are not used
    if (v.cov3d) return lr_fail(LOGRAST_ERR_ARG, "LOGRAST_BWD_ACCUMULATE_ROWS has no cov3d_precomp form");
    if (!scales || !rotations) return lr_fail(LOGRAST_ERR_ARG, "scales / rotations are NULL");
    if (reinterpret_cast<uintptr_t>(dl_dmeans3d) & 63u)
      return lr_fail(LOGRAST_ERR_ARG, "LOGRAST_BWD_ACCUMULATE_ROWS: the gradient rows must be 64-byte aligned");
    if (reinterpret_cast<uintptr_t>(rotations) & 15u) return lr_fail(LOGRAST_ERR_ARG, "rotations must be 16-byte aligned");
  } else {
    rc = lr_check_cov_args(v, scales, rotations, dl_dscales, dl_drotations);
    if (rc) return rc;
    if ((reinterpret_cast<uintptr_t>(rotations) | reinterpret_cast<uintptr_t>(dl_drotations)) & 15u)
      return lr_fail(LOGRAST_ERR_ARG, "rotations / dl_drotations must be 16-byte aligned");
  }
  if (reinterpret_cast<uintptr_t>(bwd_rows) & 63u)
    return lr_fail(LOGRAST_ERR_ARG, "bwd_rows must be 64-byte aligned (one accumulator row per line)");
  hipStream_t s = (hipStream_t)stream;
  uint32_t tiles = (uint32_t)(v.gx * v.gy);
  const uint32_t* st = reinterpret_cast<const uint32_t*>(tile_state);
  const bool accumulate = (flags & LOGRAST_BWD_ACCUMULATE) != 0 || sink_rows;
  if ((flags & LOGRAST_BWD_CONIC_TOUCHED_ONLY) && !point_weight)
    return lr_fail(LOGRAST_ERR_ARG, "LOGRAST_BWD_CONIC_TOUCHED_ONLY needs point_weight");
  // A forward with extras on a large input clears the dL/dconic rows of contributing Gaussians only (lr_stage2:
  // touched_only); the chain rule must then skip the others, which it does exactly when it is handed point_weight.
  if ((flags & LOGRAST_BWD_SCRATCH_ZEROED) && v.extras && lr_big_input(n) && !point_weight)
    return lr_fail(LOGRAST_ERR_ARG, "LOGRAST_BWD_SCRATCH_ZEROED after a forward with view.extras and n >= "
                                    "LOGRAST_HELPER_MIN_N: dL/dconic is cleared for contributing Gaussians only, pass "
                                    "point_weight (+ LOGRAST_BWD_CONIC_TOUCHED_ONLY)");
  if (!(flags & LOGRAST_BWD_SCRATCH_ZEROED))
    LR_HIP(hipMemsetAsync(dl_dconic, 0, sizeof(float) * LOGRAST_BWD_ROW_FLOATS * (size_t)n, s));
  // capacity check is a forward concern: a list that rendered is by construction within capacity
  lr_launch_blend_bwd(v, geom, st, tiles, point_list, 0xffffffffu, final_t, n_contrib, dl_dimage, dl_dconic,
                      lr_big_input(n) ? 1 : 0, v.masks, s);
  // the chain rule reads every live Gaussian's accumulator row and hands out the separate outputs: dL/dmeans2D (written
  // for all rows), dL/dopacities and dL/dcolors (written, or added to the caller's running sums)
  lr_launch_project_bwd(v, n, means3d, scales, rotations, radii, nullptr, nullptr, dl_dconic, dl_dmeans2d, dl_dopacities,
                        dl_dcolors, point_weight, dl_dmeans3d, dl_dscales, dl_drotations, accumulate, sink_rows, s);
  LR_HIP(hipGetLastError());
  return LOGRAST_OK;
}

int lograst_project_backward(const lograst_view* view, int32_t n, const float* means3d, const float* scales,
                             const float* rotations, const int32_t* radii, const float* dl_dmeans2d,
                             const float* dl_dconic, float* dl_dmeans3d, float* dl_dscales,
                             float* dl_drotations, void* stream) {
  g_prof_call++;
  LrView v;
  int rc = lr_make_view(view, &v);
  if (rc) return rc;
  if (n < 0) return lr_fail(LOGRAST_ERR_ARG, "negative Gaussian count");
  if (n == 0) return LOGRAST_OK;
  if (!means3d || !radii || !dl_dmeans2d || !dl_dconic || !dl_dmeans3d)
    return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  rc = lr_check_cov_args(v, scales, rotations, dl_dscales, dl_drotations);
  if (rc) return rc;
  if ((reinterpret_cast<uintptr_t>(dl_dconic) | reinterpret_cast<uintptr_t>(rotations) |
       reinterpret_cast<uintptr_t>(dl_drotations)) & 15u)
    return lr_fail(LOGRAST_ERR_ARG, "rotations / dl_dconic / dl_drotations must be 16-byte aligned");
  lr_launch_project_bwd(v, n, means3d, scales, rotations, radii, dl_dmeans2d, dl_dconic, nullptr, nullptr, nullptr,
                        nullptr, nullptr, dl_dmeans3d, dl_dscales, dl_drotations, false, false, (hipStream_t)stream);
  LR_HIP(hipGetLastError());
  return LOGRAST_OK;
}

size_t lograst_knn_scratch_bytes(int32_t p) { return lr_knn_scratch_bytes(p); }

int lograst_knn_mean_dist2(int32_t p, const float* points, float* out, void* scratch, size_t scratch_bytes,
                           void* stream) {
  if (p < 0) return lr_fail(LOGRAST_ERR_ARG, "negative point count");
  if (p == 0) return LOGRAST_OK;
  if (!points || !out || !scratch) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  if (scratch_bytes < lr_knn_scratch_bytes(p)) return lr_fail(LOGRAST_ERR_ARG, "knn scratch too small");
  LR_HIP(lr_launch_knn(p, points, out, scratch, scratch_bytes, (hipStream_t)stream));
  return LOGRAST_OK;
}

size_t lograst_lod_scratch_bytes(int32_t num_roots, int32_t num_nodes, int32_t max_child) {
  return lr_lod_scratch_bytes(num_roots, num_nodes, max_child > 0 ? max_child : 1);
}

int lograst_lod_traverse(int32_t num_points, int32_t num_nodes, int32_t max_child, const int32_t* node_index,
                         const int32_t* tree, const float* xyz, const float* scaling, const float* rotation,
                         const int64_t* root_index, int32_t num_roots, const float* projmatrix,
                         const float* viewmatrix, float focal_x, float focal_y, float tanfovx, float tanfovy,
                         float min_resolution_pixel, int32_t levels, int64_t* out_index, uint32_t out_capacity,
                         void* scratch, size_t scratch_bytes, void* stream) {
  if (num_points < 0 || num_nodes < 0 || num_roots < 0) return lr_fail(LOGRAST_ERR_ARG, "negative count");
  if (max_child < 1) return lr_fail(LOGRAST_ERR_ARG, "max_child must be >= 1");
  if ((uint64_t)num_nodes * (uint64_t)max_child > 0x7fffffffull) return lr_fail(LOGRAST_ERR_ARG, "tree too large");
  if (!scratch || scratch_bytes < lr_lod_scratch_bytes(num_roots, num_nodes, max_child))
    return lr_fail(LOGRAST_ERR_ARG, "lod scratch too small");
  if (num_roots > 0 && (!root_index || !node_index || !xyz || !scaling || !rotation || !projmatrix || !viewmatrix || !out_index))
    return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  if (num_nodes > 0 && !tree) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  if (levels < 0) levels = 0;
  if (levels > lr_lod_max_levels()) levels = lr_lod_max_levels();
  g_prof_call++;
  LR_HIP(lr_launch_lod(num_points, num_nodes, max_child, node_index, tree, xyz, scaling, rotation, root_index, num_roots,
                       projmatrix, viewmatrix, focal_x, focal_y, tanfovx, tanfovy, min_resolution_pixel, levels,
                       out_index, out_capacity, scratch, (hipStream_t)stream));
  return LOGRAST_OK;
}

int lograst_lod_read(const void* scratch, uint32_t* count_host, uint32_t* overflow_host, uint32_t* frontier_left_host,
                     void* stream) {
  if (!scratch || !count_host) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  uint32_t w[3] = {0, 0, 0};
  LR_HIP(hipMemcpyAsync(w, reinterpret_cast<const uint32_t*>(scratch) + lr_lod_total_word(), sizeof(w),
                        hipMemcpyDeviceToHost, (hipStream_t)stream));
  LR_HIP(hipStreamSynchronize((hipStream_t)stream));
  *count_host = w[0];
  if (overflow_host) *overflow_host = w[1];
  if (frontier_left_host) *frontier_left_host = w[2];
  return LOGRAST_OK;
}

size_t lograst_id_histogram_scratch_bytes(int32_t n) { return lr_hist_scratch_bytes(n); }

int lograst_id_histogram(int32_t n, const int32_t* point_id_pixel, int32_t num_pixels, int32_t* ids_out,
                         int64_t* counts_out, void* scratch, size_t scratch_bytes, void* stream) {
  if (n < 0 || num_pixels < 0) return lr_fail(LOGRAST_ERR_ARG, "negative count");
  if (!scratch || scratch_bytes < lr_hist_scratch_bytes(n)) return lr_fail(LOGRAST_ERR_ARG, "histogram scratch too small");
  if (n > 0 && num_pixels > 0 && (!point_id_pixel || !ids_out || !counts_out)) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  g_prof_call++;
  LR_HIP(lr_launch_id_histogram(n, point_id_pixel, num_pixels, ids_out, counts_out, scratch, (hipStream_t)stream));
  return LOGRAST_OK;
}

int lograst_id_histogram_read(const void* scratch, uint32_t* count_host, void* stream) {
  if (!scratch || !count_host) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  LR_HIP(hipMemcpyAsync(count_host, scratch, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  LR_HIP(hipStreamSynchronize((hipStream_t)stream));
  return LOGRAST_OK;
}

int lograst_counter_update(int32_t nv, const int64_t* visible_index, const float* grad_means2d, const int32_t* radii,
                           const float* point_weight, int32_t k, const int32_t* point_id, const int64_t* point_count,
                           int32_t num_points, float* weights_max, float* weights_sum, float* grad_sum,
                           int16_t* radii_max, int16_t* visible_count, int32_t* radii_max_max, int32_t* area_sum,
                           int32_t* create_steps, uint8_t* flag_vis_out, void* stream) {
  if (nv < 0 || k < 0 || num_points < 0) return lr_fail(LOGRAST_ERR_ARG, "negative count");
  if (nv == 0) return LOGRAST_OK;
  if (!visible_index || !grad_means2d || !radii || !point_weight || (k > 0 && (!point_id || !point_count)))
    return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  if (!weights_max || !weights_sum || !grad_sum || !radii_max || !visible_count || !radii_max_max || !area_sum || !create_steps)
    return lr_fail(LOGRAST_ERR_ARG, "NULL counter buffer");
  CounterArgs a;
  a.visible_index = visible_index; a.grad = grad_means2d; a.radii = radii; a.weight = point_weight;
  a.point_id = point_id; a.point_count = point_count;
  a.weights_max = weights_max; a.weights_sum = weights_sum; a.grad_sum = grad_sum;
  a.radii_max = radii_max; a.visible_count = visible_count;
  a.radii_max_max = radii_max_max; a.area_sum = area_sum; a.create_steps = create_steps;
  a.flag_vis = flag_vis_out;
  a.nv = nv; a.k = k; a.num_points = num_points;
  g_prof_call++;
  LR_HIP(lr_launch_counter(a, (hipStream_t)stream));
  return LOGRAST_OK;
}

int lograst_sparse_adam(int32_t m, int32_t num_points, const int64_t* index, const uint8_t* flag_vis,
                        int32_t num_keys, const lograst_adam_key* keys, double beta1, double beta2,
                        double bias_correction2_sqrt, double eps, void* stream) {
  if (m < 0 || num_points < 0) return lr_fail(LOGRAST_ERR_ARG, "negative count");
  if (num_keys < 0 || num_keys > ADAM_MAX_KEYS) return lr_fail(LOGRAST_ERR_ARG, "at most 8 keys per call");
  if (m == 0 || num_keys == 0) return LOGRAST_OK;
  if (!index || !flag_vis || !keys) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  AdamArgs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < num_keys; i++) {
    const lograst_adam_key& k = keys[i];
    if (!k.model_param || !k.param || !k.grad || !k.exp_avg || !k.exp_avg_sq) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer in key");
    if (k.width < 1) return lr_fail(LOGRAST_ERR_ARG, "key width must be >= 1");
    a.key[i].model = (float*)k.model_param; a.key[i].param = (const float*)k.param; a.key[i].grad = (const float*)k.grad;
    a.key[i].exp_avg = (float*)k.exp_avg; a.key[i].exp_avg_sq = (float*)k.exp_avg_sq;
    a.key[i].max_exp_avg_sq = (float*)k.max_exp_avg_sq;
    a.key[i].width = k.width; a.key[i].neg_step_size = -k.step_size;
  }
  a.index = index; a.flag_vis = flag_vis; a.m = m; a.num_points = num_points;
  // the reference's Python scalars are doubles that torch narrows to fp32 when they meet an fp32 tensor
  a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
  a.bc2_sqrt = (float)bias_correction2_sqrt; a.eps = (float)eps;
  g_prof_call++;
  LR_HIP(lr_launch_sparse_adam(a, num_keys, (hipStream_t)stream));
  return LOGRAST_OK;
}

static int lr_ga_check(int32_t n, int32_t sh_coeffs, int32_t active_degree, const float* camera_center) {
  if (n < 0) return lr_fail(LOGRAST_ERR_ARG, "negative row count");
  if (sh_coeffs < 0 || sh_coeffs > 15) return lr_fail(LOGRAST_ERR_ARG, "sh_coeffs must be 0..15 (degree <= 3)");
  if (active_degree < 0 || active_degree > 3) return lr_fail(LOGRAST_ERR_ARG, "active SH degree must be 0..3");
  if (active_degree > 0 && (active_degree + 1) * (active_degree + 1) - 1 > sh_coeffs)
    return lr_fail(LOGRAST_ERR_ARG, "shs holds fewer coefficients than the active degree needs");
  if (active_degree > 0 && !camera_center) return lr_fail(LOGRAST_ERR_ARG, "camera_center is NULL");
  return LOGRAST_OK;
}

int lograst_gather_activate(int32_t n, int32_t num_points, const int64_t* index, const float* xyz,
                            const float* scaling, const float* opacity, const float* rotation, const float* colors,
                            const float* shs, int32_t sh_coeffs, int32_t active_degree, const float* camera_center,
                            float* raw_xyz, float* raw_scaling, float* raw_opacity, float* raw_rotation,
                            float* raw_colors, float* raw_shs, float* act_scaling, float* act_opacity,
                            float* act_rotation, float* act_colors, void* stream) {
  int rc = lr_ga_check(n, sh_coeffs, active_degree, camera_center);
  if (rc) return rc;
  if (n == 0) return LOGRAST_OK;
  if (num_points <= 0) return lr_fail(LOGRAST_ERR_ARG, "rows requested from an empty model");
  if (!index || !xyz || !scaling || !opacity || !rotation || !colors || !raw_xyz || !raw_scaling || !raw_opacity ||
      !raw_rotation || !raw_colors || !act_scaling || !act_opacity || !act_rotation || !act_colors)
    return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  if (sh_coeffs > 0 && (!shs || !raw_shs)) return lr_fail(LOGRAST_ERR_ARG, "NULL shs pointer");
  GatherArgs a;
  a.index = index; a.xyz = xyz; a.scaling = scaling; a.opacity = opacity; a.rotation = rotation; a.colors = colors;
  a.shs = shs; a.campos = camera_center;
  a.r_xyz = raw_xyz; a.r_scaling = raw_scaling; a.r_opacity = raw_opacity; a.r_rotation = raw_rotation;
  a.r_colors = raw_colors; a.r_shs = raw_shs;
  a.a_scaling = act_scaling; a.a_opacity = act_opacity; a.a_rotation = act_rotation; a.a_colors = act_colors;
  a.n = n; a.num_points = num_points; a.K = sh_coeffs; a.deg = active_degree;
  g_prof_call++;
  LR_HIP(lr_launch_gather_activate(a, (hipStream_t)stream));
  return LOGRAST_OK;
}

int lograst_activate_backward(int32_t n, const float* raw_xyz, const float* raw_scaling, const float* raw_opacity,
                              const float* raw_rotation, int32_t sh_coeffs, int32_t active_degree,
                              const float* camera_center, const float* dl_dact_scaling, const float* dl_dact_opacity,
                              const float* dl_dact_rotation, const float* dl_dact_colors, float* dl_dscaling,
                              float* dl_dopacity, float* dl_drotation, float* dl_dcolors, float* dl_dshs, void* stream) {
  int rc = lr_ga_check(n, sh_coeffs, active_degree, camera_center);
  if (rc) return rc;
  if (n == 0) return LOGRAST_OK;
  if (!raw_xyz || !raw_scaling || !raw_opacity || !raw_rotation || !dl_dact_scaling || !dl_dact_opacity ||
      !dl_dact_rotation || !dl_dact_colors || !dl_dscaling || !dl_dopacity || !dl_drotation || !dl_dcolors)
    return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  ActBwdArgs a;
  a.r_xyz = raw_xyz; a.r_scaling = raw_scaling; a.r_opacity = raw_opacity; a.r_rotation = raw_rotation;
  a.campos = camera_center;
  a.g_a_scaling = dl_dact_scaling; a.g_a_opacity = dl_dact_opacity; a.g_a_rotation = dl_dact_rotation;
  a.g_a_colors = dl_dact_colors;
  a.g_scaling = dl_dscaling; a.g_opacity = dl_dopacity; a.g_rotation = dl_drotation; a.g_colors = dl_dcolors;
  a.g_shs = dl_dshs;
  a.n = n; a.K = sh_coeffs; a.deg = active_degree;
  g_prof_call++;
  LR_HIP(lr_launch_activate_bwd(a, (hipStream_t)stream));
  return LOGRAST_OK;
}

int lograst_activate_backward_adam(int32_t n, const float* raw_xyz, const float* raw_scaling, const float* raw_opacity,
                                   const float* raw_rotation, int32_t sh_coeffs, int32_t active_degree,
                                   const float* camera_center, const float* dl_dact_xyz, const float* dl_dact_scaling,
                                   const float* dl_dact_opacity, const float* dl_dact_rotation, const float* dl_dact_colors,
                                   int32_t num_points, const int64_t* index, const int32_t* radii,
                                   const lograst_adam_key* keys, double beta1, double beta2, double bias_correction2_sqrt,
                                   double eps, void* stream) {
  int rc = lr_ga_check(n, sh_coeffs, active_degree, camera_center);
  if (rc) return rc;
  if (n == 0) return LOGRAST_OK;
  if (num_points <= 0) return lr_fail(LOGRAST_ERR_ARG, "rows of an empty model");
  if (!raw_xyz || !raw_scaling || !raw_opacity || !raw_rotation || !dl_dact_xyz || !dl_dact_scaling || !dl_dact_opacity ||
      !dl_dact_rotation || !dl_dact_colors || !index || !radii || !keys)
    return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  static const int widths[6] = {3, 3, 1, 4, 3, 0};
  ActBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.r_xyz = raw_xyz; a.r_scaling = raw_scaling; a.r_opacity = raw_opacity; a.r_rotation = raw_rotation;
  a.campos = camera_center;
  a.g_a_scaling = dl_dact_scaling; a.g_a_opacity = dl_dact_opacity; a.g_a_rotation = dl_dact_rotation;
  a.g_a_colors = dl_dact_colors;
  a.n = n; a.K = sh_coeffs; a.deg = active_degree;
  AdamArgs f;
  memset(&f, 0, sizeof(f));
  for (int i = 0; i < 6; i++) {
    const lograst_adam_key& k = keys[i];
    if (!k.model_param) continue;                            // key not optimised in this step
    const int w = i == 5 ? 3 * sh_coeffs : widths[i];
    if (k.width != w) return lr_fail(LOGRAST_ERR_ARG, "lograst_activate_backward_adam: key widths are 3, 3, 1, 4, 3, 3 * sh_coeffs (xyz, scaling, opacity, rotation, colors, shs)");
    if (i == 5 && (sh_coeffs == 0 || active_degree == 0)) return lr_fail(LOGRAST_ERR_ARG, "shs key without active SH coefficients");
    if (!k.param || !k.exp_avg || !k.exp_avg_sq) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer in key");
    a.g_shs = nullptr;
    f.key[i].model = (float*)k.model_param; f.key[i].param = (const float*)k.param; f.key[i].grad = nullptr;
    f.key[i].exp_avg = (float*)k.exp_avg; f.key[i].exp_avg_sq = (float*)k.exp_avg_sq;
    f.key[i].max_exp_avg_sq = (float*)k.max_exp_avg_sq;
    f.key[i].width = k.width; f.key[i].neg_step_size = -k.step_size;
  }
  if ((reinterpret_cast<uintptr_t>(raw_rotation) | reinterpret_cast<uintptr_t>(dl_dact_rotation)) & 15u)
    return lr_fail(LOGRAST_ERR_ARG, "raw_rotation / dl_dact_rotation must be 16-byte aligned");
  f.index = index; f.flag_vis = nullptr; f.m = n; f.num_points = num_points;
  f.beta1 = (float)beta1; f.beta2 = (float)beta2; f.omb1 = (float)(1.0 - beta1); f.omb2 = (float)(1.0 - beta2);
  f.bc2_sqrt = (float)bias_correction2_sqrt; f.eps = (float)eps;
  g_prof_call++;
  LR_HIP(lr_launch_activate_bwd_adam(a, f, dl_dact_xyz, radii, (hipStream_t)stream));
  return LOGRAST_OK;
}

static int lr_sh_check(int32_t n, int32_t degree, int32_t max_coeffs) {
  if (n < 0) return lr_fail(LOGRAST_ERR_ARG, "negative Gaussian count");
  if (degree < 0 || degree > 3) return lr_fail(LOGRAST_ERR_ARG, "SH degree must be 0..3");
  if (max_coeffs < (degree + 1) * (degree + 1) || max_coeffs > 16)
    return lr_fail(LOGRAST_ERR_ARG, "shs must hold (degree+1)^2 .. 16 coefficients per Gaussian");
  return LOGRAST_OK;
}

int lograst_sh_forward(int32_t n, int32_t degree, int32_t max_coeffs, const float* means3d, const float* campos,
                       const float* shs, float* colors, uint8_t* clamped, void* stream) {
  int rc = lr_sh_check(n, degree, max_coeffs);
  if (rc) return rc;
  if (n == 0) return LOGRAST_OK;
  if (!means3d || !campos || !shs || !colors || !clamped) return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  lr_launch_sh_fwd(n, degree, max_coeffs, means3d, campos, shs, colors, clamped, (hipStream_t)stream);
  LR_HIP(hipGetLastError());
  return LOGRAST_OK;
}

int lograst_sh_backward(int32_t n, int32_t degree, int32_t max_coeffs, const float* means3d, const float* campos,
                        const float* shs, const uint8_t* clamped, const float* dl_dcolors, float* dl_dshs,
                        float* dl_dmeans3d, int32_t accumulate, void* stream) {
  int rc = lr_sh_check(n, degree, max_coeffs);
  if (rc) return rc;
  if (n == 0) return LOGRAST_OK;
  if (!means3d || !campos || !shs || !clamped || !dl_dcolors || !dl_dshs || !dl_dmeans3d)
    return lr_fail(LOGRAST_ERR_ARG, "NULL pointer");
  lr_launch_sh_bwd(n, degree, max_coeffs, means3d, campos, shs, clamped, dl_dcolors, dl_dshs, dl_dmeans3d,
                   accumulate != 0, (hipStream_t)stream);
  LR_HIP(hipGetLastError());
  return LOGRAST_OK;
}

void lograst_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
}
void lograst_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  lr_prof_drain_locked();
  std::memset(g_prof_ms, 0, sizeof(g_prof_ms));
  std::memset(g_prof_cnt, 0, sizeof(g_prof_cnt));
}
int lograst_profile_read(double* ms_out, int64_t* count_out) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  lr_prof_drain_locked();
  for (int i = 0; i < LOGRAST_NUM_KERNELS; i++) {
    if (ms_out) ms_out[i] = g_prof_ms[i];
    if (count_out) count_out[i] = g_prof_cnt[i];
  }
  return LOGRAST_OK;
}
const char* lograst_kernel_name(int slot) {
  return (slot >= 0 && slot < LOGRAST_NUM_KERNELS) ? kKernelNames[slot] : "";
}

}  // extern "C"

/*
 * lograst.h -- C ABI of liblograst.so: the MI355X (gfx950) differentiable Gaussian-splatting
 * rasterizer behind LoG's operator surface.
 *
 * The reference (zju3dv/LoG) has no C ABI of its own: its boundary is two pybind11 torch extensions
 * built from third-party repos plus one in-tree JIT extension.  Each entry point below
 * names the reference interface it stands behind; the Python shims that reproduce those interfaces
 * one-to-one live in log_amd/rasterizer.py and log_amd/compute_radius.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; all floats are fp32;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); nothing synchronises the
 *     host except lograst_forward_project when `num_instances_host` != NULL, and lograst_profile_read;
 *   - no allocation happens inside the library: the caller sizes scratch with the *_bytes helpers
 *     (the Python shim uses torch's caching allocator);
 *   - return value 0 = success, negative = error (lograst_last_error() gives the text);
 *   - matrices use LoG's row-vector convention, flat row-major: t = p_row @ M
 *     (/root/reference/LoG/dataset/base.py:40-46).
 */
#ifndef LOGRAST_H
#define LOGRAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOGRAST_VERSION 4   /* 4 (round 6): contracts that changed since 3 -- (a) lograst_geom_bytes grew by 8 B per Gaussian + 256 KB (the
 * rank rows of the 5-16-tile rects behind the fill records and indices: a caller that sized `geom` as 84 B per Gaussian gets
 * out-of-bounds writes; always size it with lograst_geom_bytes); (b) bwd_rows slots 12-15 are scratch of the chain rule
 * (band views keep the compact live-row list there); (c) point_list tails of streamed lists (> 4096 keys) are unspecified
 * until lograst_finish_lists, which now checks that `keys` is the buffer the forward filled and synchronises the stream;
 * (d) lograst_view gained hit_masks / hit_mask_words / hit_mask_form: an optional buffer in which the forward leaves its
 * per-chunk support ballots for the reverse walk (below); new entry points: lograst_hit_mask_bytes, lograst_forward_form,
 * lograst_pack_rows_clear, lograst_unpack_rows(atomic = 2), lograst_activate_backward_adam.  Added later in round 6 without a
 * version change (new entry points only): lograst_pack_rows_hinted, lograst_add_visible, lograst_add_visible_n.
 * 2: lograst_view gained cov3d_precomp / dl_dcov3d; 3: the backward accumulates into 64-byte rows (bwd_rows); lograst_view gained walk_form.  Added since without a version change (new entry points only): lograst_sparse_segment_floats / lograst_pack_rows / lograst_unpack_rows / lograst_ordered_lengths / lograst_finish_lists */
#define LOGRAST_TILE 16        /* pixels per tile side (tile rects are part of the integer contract) */
#define LOGRAST_REC_FLOATS 16  /* floats per projected-Gaussian record (64 B): see log_amd/csrc/project.hip */
/* The reverse walk's accumulators: ONE 64-byte row per Gaussian -- slots 0-1 dL/d(ndc mean x, y), 2-4 dL/d(conic A, B, C),
 * 5 dL/dopacity, 6-8 dL/dcolour r g b, 9-11 unused, 12-15 scratch of the chain rule (band views keep the compact list of
 * the live rows there: row 0 its length, rows 1.. four indices each; whatever the caller finds in bwd_rows after
 * lograst_backward is unspecified).  A memory-side atomic costs one operation per 64-byte line whatever the number of lanes
 * in it (tools/micro/atomic_lines.hip), so a (wave, Gaussian) visit commits all nine sums as one. */
#define LOGRAST_BWD_ROW_FLOATS 16

/* 2-D low-pass flavours */
#define LOGRAST_FILTER_NONE 0   /* use_filter=False of the fork (LoG/render/renderer.py:151-152) */
#define LOGRAST_FORM_AUTO 0
#define LOGRAST_FORM_ROWS 1
#define LOGRAST_FORM_QUADRANT 2
#define LOGRAST_FILTER_DILATE 1 /* upstream package: cov.xx += 0.3 (LoG/model/geometry.py:87-88) */
#define LOGRAST_FILTER_CLAMP 2  /* `wodilate` fork: cov.xx = max(cov.xx, 0.3) (LoG/cuda/compute_radius_kernel.cu:100-104) */

/* Caller-owned status block of the forward (device words; see lograst_forward_render) */
#define LOGRAST_STATUS_WORDS 8
#define LOGRAST_STATUS_OVERFLOW 0        /* sticky: bit 0 set by every forward that overflowed */
#define LOGRAST_STATUS_LAST_INSTANCES 1  /* the most recent forward: tile instances, ... */
#define LOGRAST_STATUS_LAST_OVERFLOW 2   /* ... whether it overflowed, ... */
#define LOGRAST_STATUS_LAST_MAX_LEN 3    /* ... its longest tile list, ... */
#define LOGRAST_STATUS_LAST_RECT 4       /* ... and its rect-rule instance count */
#define LOGRAST_STATUS_MAX_INSTANCES 5   /* running maxima over all forwards since the caller cleared the block */
#define LOGRAST_STATUS_MAX_MAX_LEN 6
#define LOGRAST_STATUS_FORWARDS 7        /* forwards recorded since then */

/* error codes */
#define LOGRAST_OK 0
#define LOGRAST_ERR_ARG -1
#define LOGRAST_ERR_HIP -2

/* Per-call view description.  Stands for GaussianRasterizationSettings
 * (/root/reference/LoG/render/renderer.py:63-76): image_height/width, tanfovx/y, bg, scale_modifier,
 * viewmatrix, projmatrix.  sh_degree/campos/prefiltered/debug have no effect on this path (LoG always
 * passes colors_precomp, renderer.py:72-75,144-145). */
typedef struct lograst_view {
  int32_t width, height;
  float tanfovx, tanfovy;
  float scale_modifier;
  int32_t filter_mode;     /* LOGRAST_FILTER_* */
  int32_t ndc_cull;        /* 1: also cull |ndc.x|,|ndc.y| > 1.3 (compute_radius_kernel.cu:131-134) */
  int32_t extras;          /* 1: produce the fork's point_id_pixel / point_weight_pixel / point_weight */
  const float* viewmatrix; /* device, 16 floats */
  const float* projmatrix; /* device, 16 floats */
  const float* bg;         /* device, 3 floats */
  /* Image split across GPUs ("per-GPU tile ownership"; new design, the reference is single-GPU): only
   * the tile rows [tile_row_begin, tile_row_end) are rendered -- every Gaussian's rect is clipped to them, so lists,
   * image rows, gradients and point_weight cover this band only, radii is 0 for Gaussians that do not reach it, and
   * pixels outside the band come out as background.  Both 0 = the whole image. */
  int32_t tile_row_begin, tile_row_end;
  /* The rasterizer's `cov3D_precomp` argument (the third-party forward's alternative to scales + rotations; LoG itself
   * never passes it, /root/reference/LoG/render/renderer.py:134,149): device, n x 6 floats, the upper triangle
   * (xx, xy, xz, yy, yz, zz) of every Gaussian's world-space covariance, or NULL.  When set, `scales` / `rotations` are
   * not read (they may be NULL) and scale_modifier has no effect (it scales the `scales` only); the backward entry points
   * then write dL/dcov3D (n x 6, off-diagonal entries carry both symmetric positions: 2 x the matrix partial) to
   * `dl_dcov3d` and leave dl_dscales / dl_drotations (may be NULL) alone. */
  const float* cov3d_precomp;
  float* dl_dcov3d;
  /* Which form of the compositing kernels (speed only, never a result; knobs LOGRAST_FWD_ROWS / LOGRAST_BWD_ROWS = 2
   * follow it, 0 / 1 override it): LOGRAST_FORM_AUTO = the library decides from n alone; LOGRAST_FORM_ROWS = the wave's
   * four 16-lane rows walk their own 4x4 pixel blocks (views of tiny splats: few tile instances per Gaussian);
   * LOGRAST_FORM_QUADRANT = the whole wave walks its 8x8 quadrant.  The caller knows the view's instances per Gaussian
   * from earlier forwards (forward) or from this view's forward (backward). */
  int32_t walk_form;
  /* Optional (version 4; NULL = off): a buffer in which the forward's compositing kernels leave, per wave and 64-entry chunk
   * of a tile list they walked, the ballot of their support tests, and from which the reverse walk of lograst_backward
   * takes its visits instead of running the tests (and gathering every record of a chunk) again -- same decisions, same
   * sums.  Device, 32-byte aligned, hit_mask_words 64-bit words >= lograst_hit_mask_bytes(capacity, width, height) / 8,
   * uninitialised; the backward must be given the very buffer (contents untouched) of the forward whose tile_state it is
   * handed.  Speed only: 30 M Gaussians, reverse walk 640 -> see DESIGN.md section 4. */
  uint64_t* hit_masks;
  uint64_t hit_mask_words;
  /* lograst_backward only: what lograst_forward_form() returned for the forward's view (1 row-split, 2 quadrant; 0 = do not
   * use the masks).  The reverse walk takes the masks only when it runs in that same form (walk_form / LOGRAST_BWD_ROWS). */
  int32_t hit_mask_form;
} lograst_view;

int lograst_version(void);
const char* lograst_last_error(void);

/* ---- sizing helpers --------------------------------------------------------------------------- */
/* bytes of the per-tile state block for a WxH image and n Gaussians (header, counters, offsets, cursors, dispatch
 * order, and the per-batch slot reservations of the projection stage, which grow with n) */
size_t lograst_tile_state_bytes(int32_t width, int32_t height, int32_t n);
/* bytes of the projected-record array for N Gaussians (64-byte records followed by 16-byte fill records and a 4-byte
 * index each, then -- round 5 -- the rank rows of the rects of 5..16 tiles: 32 bytes per such rect, room for one in four
 * Gaussians of every projection batch: 8 bytes per Gaussian + 256 KB).  Records of Gaussians with radii == 0 are
 * undefined: a view that owns a band of tile rows does not write them (a Gaussian without a rect then costs the 40 bytes
 * its rect is computed from and its radii word). */
size_t lograst_geom_bytes(int32_t n);
/* bytes of the (depth,id) key buffer (keys + an equally large scratch half used by the long-list sort) / of the
 * sorted id list, for `capacity` tile instances */
size_t lograst_keys_bytes(uint32_t capacity);
/* bytes of lograst_view.hit_masks for `capacity` tile instances on a width x height image: 128 per (tile, 64-entry chunk)
 * slot, capacity / 64 + tiles + 1 slots (only the chunks a view walks are ever written or read) */
size_t lograst_hit_mask_bytes(uint32_t capacity, int32_t width, int32_t height);
/* which form the forward's compositing kernels take for this view (its walk_form + the LOGRAST_FWD_ROWS knob): 1 = row-split,
 * 2 = quadrant, negative = error.  Pass it back as hit_mask_form of the backward's view. */
int lograst_forward_form(const lograst_view* view);
size_t lograst_list_bytes(uint32_t capacity);

/* ---- A0: LoG/cuda compute_radius --------------------------------------------------------------
 * Replaces compute_radius_module.compute_radius (/root/reference/LoG/cuda/compute_radius_kernel.cu:158-187;
 * caller LoG/model/level_of_gaussian.py:81-84): projected 3-sigma radius in pixels (float, no ceil),
 * 0 where |ndc| > 1.3 or det == 0. */
int lograst_compute_radius(int32_t p, const float* means3d, const float* scales, const float* rotations,
                           const float* projmatrix, const float* viewmatrix, float focal_x, float focal_y,
                           float tanfovx, float tanfovy, float* radii_out, void* stream);

/* ---- forward, stage 1: projection + tile counting + scan (A1, A2) -----------------------------
 * Stands for the first half of GaussianRasterizer.forward (called at LoG/render/renderer.py:153,190-198;
 * LoG/model/level_of_gaussian.py:211-219).  Writes radii[n] (API output), geom (n records), and
 * tile_state (per-tile counts/offsets).  The total number of tile instances is left in
 * tile_state and, if num_instances_host != NULL (pinned or pageable host memory), also copied there
 * after a stream synchronise so the caller can size the key/list buffers exactly; max_tile_len_host (optional)
 * receives the longest tile list, which stage 2 uses to launch only the sort levels that are needed.
 * geom must hold lograst_geom_bytes(n) and tile_state lograst_tile_state_bytes(width, height, n) bytes for THIS n
 * (both grow with n: per-Gaussian fill records, per-batch slot reservations). */
int lograst_forward_project(const lograst_view* view, int32_t n, const float* means3d, const float* scales,
                            const float* rotations, const float* opacities, const float* colors,
                            int32_t* radii, void* geom, void* tile_state, uint32_t* num_instances_host,
                            uint32_t* max_tile_len_host, void* stream);

/* ---- forward, stage 2: per-tile bucketing + per-tile depth sort + compositing (A3, A4, A5, A7, A8)
 * keys: scratch of lograst_keys_bytes(capacity) (dead after the call -- unless lograst_finish_lists is to complete
 * lazily ordered lists, see there); point_list: lograst_list_bytes(capacity), kept for backward: the ids of every tile
 * list in (depth, id) order as far as the view's walk needed them (all of a list of up to 4096 keys; at least the first
 * window of a longer one: lograst_ordered_lengths).  `capacity` = number of tile instances the two buffers can hold.  With the
 * exact count from stage 1 it always suffices.  With a guess (sync-free operation) the kernels never write past
 * it: if the real count is larger NOTHING is rendered, the overflow flag in tile_state is raised, and the
 * caller finds out from lograst_read_state() (the call itself cannot know without a host sync).
 * max_tile_len: host-side upper bound on the longest tile list (from stage 1, or a hint; 0 = unknown, treated
 * as `capacity`): lists longer than 8192 keys take a multi-pass sort whose number of launches depends on it.  An
 * under-estimate leaves such lists partially sorted, so pass 0 when in doubt.
 * Outputs: image[3,H,W], final_T[H,W], n_contrib[H,W] (both kept for backward), and when
 * view->extras: point_id_pixel[H,W] (i32, -1 = none), point_weight_pixel[H,W], point_weight[n].
 * bwd_scratch (optional, NULL/0 = none; bwd_scratch_floats must be 0 or LOGRAST_BWD_ROW_FLOATS): the n x 16 fp32
 * accumulator rows of lograst_backward (its `bwd_rows`), zero-filled here (inside a kernel this call launches anyway) for
 * a caller that will differentiate this view: pass the block to lograst_backward with LOGRAST_BWD_SCRATCH_ZEROED -- no
 * separate memset.  With view->extras on large inputs (n >= 4,000,000, env LOGRAST_HELPER_MIN_N) only the rows of
 * Gaussians that contributed to a pixel are cleared (point_weight > 0; the compositing kernel clears a row when it meets
 * the Gaussian; below that size the whole block is cleared): always pass point_weight and
 * LOGRAST_BWD_CONIC_TOUCHED_ONLY to lograst_backward after a forward with view->extras (correct at every size;
 * lograst_backward rejects LOGRAST_BWD_SCRATCH_ZEROED without point_weight on large inputs).
 * max_tile_len is also CHECKED on the device: when the real longest list exceeds a non-zero max_tile_len (or the
 * instance count exceeds capacity) nothing is sorted or composited and the overflow flag of tile_state is raised; the
 * outputs of such a call are undefined.  status (optional): LOGRAST_STATUS_WORDS device words owned by the caller
 * (zeroed once), into which every forward records itself -- see LOGRAST_STATUS_* -- so that a caller running many
 * sync-free forwards, on any number of streams, checks all of them with one read-back.
 * bwd_scratch must be 64-byte aligned. */
int lograst_forward_render(const lograst_view* view, int32_t n, const void* geom, void* tile_state,
                           uint64_t* keys, uint32_t* point_list, uint32_t capacity, uint32_t max_tile_len,
                           float* image, float* final_t, int32_t* n_contrib, int32_t* point_id_pixel,
                           float* point_weight_pixel, float* point_weight, float* bwd_scratch,
                           int32_t bwd_scratch_floats, uint32_t* status, void* stream);

/* ---- forward in ONE call (sync-free operation) ---------------------------------------------------------------
 * lograst_forward_project + lograst_forward_render back to back for a caller that already knows a capacity (and a
 * max_tile_len, or 0): one boundary crossing and no host decision between the stages -- what GaussianRasterizer.forward
 * (LoG/render/renderer.py:153) costs the host is then one call.  Arguments as in the two stage calls. */
int lograst_forward(const lograst_view* view, int32_t n, const float* means3d, const float* scales,
                    const float* rotations, const float* opacities, const float* colors, int32_t* radii, void* geom,
                    void* tile_state, uint64_t* keys, uint32_t* point_list, uint32_t capacity, uint32_t max_tile_len,
                    float* image, float* final_t, int32_t* n_contrib, int32_t* point_id_pixel,
                    float* point_weight_pixel, float* point_weight, float* bwd_scratch, int32_t bwd_scratch_floats,
                    uint32_t* status, void* stream);

/* The forward in one call for a caller that has only a GUESS of the capacity (and of max_tile_len): what the drop-in
 * package's default mode runs (log_amd/rasterizer.py).  Stage 1, then stage 2 with the guessed buffers are enqueued back
 * to back -- the stream never waits for the host --, while the stage-1 header {instance count, longest list} is copied
 * to the host on an internal side stream that waits for stage 1 only; the call returns once that copy has landed
 * (stage 2 is then still running or queued).  If *num_instances_host <= capacity and (max_tile_len == 0 or
 * *max_tile_len_host <= max_tile_len) the speculative stage 2 is the forward.  Otherwise its kernels returned without
 * rendering (as in lograst_forward_render with too small a capacity) and the caller repeats stage 2 alone with exact
 * buffers: lograst_forward_render(view, n, geom, tile_state, <keys / point_list for *num_instances_host>, ...).  An
 * attempt that overflows does not touch `status` (the repeat records the forward).  Synchronises the side stream, not
 * `stream`; not capturable into a HIP graph (use lograst_forward with a known capacity there).  Stands for the
 * num_rendered read-back of the third-party forward (called at LoG/render/renderer.py:153) without its pipeline bubble. */
int lograst_forward_speculative(const lograst_view* view, int32_t n, const float* means3d, const float* scales,
                                const float* rotations, const float* opacities, const float* colors, int32_t* radii,
                                void* geom, void* tile_state, uint64_t* keys, uint32_t* point_list, uint32_t capacity,
                                uint32_t max_tile_len, float* image, float* final_t, int32_t* n_contrib,
                                int32_t* point_id_pixel, float* point_weight_pixel, float* point_weight,
                                float* bwd_scratch, int32_t bwd_scratch_floats, uint32_t* status,
                                uint32_t* num_instances_host, uint32_t* max_tile_len_host, void* stream);

/* Image split into bands of tile rows (new design, SURVEY 8e / BASELINE configs[4]; view->tile_row_begin/end): the
 * tile-row range every Gaussian's rect covers on the WHOLE image, rows_out[i] = y0 | y1 << 16 (rows [y0, y1); 0 = the
 * projection drops this Gaussian), computed by the projection's own code: the Gaussians lograst_forward keeps for a band
 * [b, e) are exactly those with y0 < e and y1 > b.  A rank that owns a band renders from that subset only (same relative
 * order => same tile lists, bit-identical band).  view->tile_row_begin/end are ignored here. */
int lograst_tile_rows(const lograst_view* view, int32_t n, const float* means3d, const float* scales,
                      const float* rotations, uint32_t* rows_out, void* stream);

/* Measurement helper (bench.py's HBM denominator, SURVEY 8d "measured device-copy bandwidth"): streams `bytes` from
 * src to dst with 16-byte accesses.  `blocks` & 0xfffff = workgroups of 256 of the grid-stride forms (0: 4096);
 * `blocks` >> 20 = form: 0 grid-stride, four non-temporal loads in flight per lane, non-temporal stores; 1 one access
 * per lane, no loop, plain loads / stores; 2 as 0 with eight loads in flight; 3 as 0 with plain loads; 4 as 1,
 * non-temporal.  Pointers and size must be multiples of 16 bytes.  Not on the rasterizer's path. */
int lograst_stream_copy(void* dst, const void* src, size_t bytes, int32_t blocks, void* stream);

/* ---- row-sparse gradient exchange: pack / unpack (log_amd/dist.py; the reference has no multi-GPU code: new surface) ---
 * rows: float32 [groups][rows_per_group][16] -- a row-major running-sum bucket (LOGRAST_GRAD_ROW_FLOATS = 16), 64-byte
 * aligned.  lograst_pack_rows writes one SEGMENT per group, back to back, each lograst_sparse_segment_floats(kmax) floats:
 * header [16] (word 0: how many rows with a non-zero entry the group holds), values [kmax][16] (those rows, in no
 * particular order), index [roundup(kmax, 16)] (int32: the row's index inside its group).  Rows beyond kmax are dropped
 * and *overflow (a device word, OR-ed; may be NULL) is raised.  Equal-sized segments: an all-to-all / all-gather with
 * equal splits moves them.  lograst_unpack_rows: for every segment s and every row j < min(count_s, kmax):
 *   atomic != 0:  dest[index][0..15] += values            (all segments into the same rows_per_group rows.  Despite the
 *                 argument's name no float atomics are involved since round 5: the segments are added one after the
 *                 other in segment order -- rows inside one segment are unique --, so a row's sum is
 *                 ((s0 + s1) + s2) + ... on every run, whatever order the segments arrived in)
 *   atomic == 0:  dest[s * dest_group_rows + index][..] = values   (segment s owns its own range of rows: plain stores;
 *                 dest_group_rows >= rows_per_group)
 *   atomic == 2:  dest[s * dest_group_rows + index][..] = 0        (version 4: clears exactly the rows an earlier
 *                 atomic == 0 call with the same segments wrote -- cheaper than zero-filling a mostly empty result)
 * dest and packed must be 16-byte aligned. */
size_t lograst_sparse_segment_floats(int32_t kmax);
int lograst_pack_rows(const float* rows, int32_t groups, int64_t rows_per_group, int32_t kmax, float* packed,
                      uint32_t* overflow, void* stream);
/* lograst_pack_rows that also ZEROES every row it packed in `rows` ("pack and clear", version 4): the bucket of a group of
 * views is all zero again once its rows are on their way, so a step that streams its exchange group by group
 * (log_amd.dist.StepExchange, parts > 1, sparse) never zero-fills its buckets.  Rows dropped by an exceeded kmax stay. */
int lograst_pack_rows_clear(float* rows, int32_t groups, int64_t rows_per_group, int32_t kmax, float* packed,
                            uint32_t* overflow, void* stream);
/* The same with a HINT (added late in round 6, new entry point only): `hint` holds one 32-bit word per row of the whole array
 * (row g * rows_per_group + r; rows from hint_rows on have none).  A row whose word is zero is, by the caller's contract, all
 * zero: it is neither read nor packed (nor cleared); a row whose word is non-zero is packed whatever it holds (an all-zero
 * row then travels as zeros).  The word is compared as bits: a view's point_weight [n] (float, zero
 * exactly for the Gaussians that contributed to no pixel -- lograst_backward leaves their gradient rows untouched) is such a
 * hint for a bucket that holds that ONE view's gradient rows: the scan then reads 4 bytes per row instead of 64.  clear != 0:
 * as lograst_pack_rows_clear. */
int lograst_pack_rows_hinted(float* rows, int32_t groups, int64_t rows_per_group, int32_t kmax, float* packed,
                             uint32_t* overflow, int32_t clear, const uint32_t* hint, int64_t hint_rows, void* stream);
int lograst_unpack_rows(float* dest, const float* packed, int32_t segments, int32_t kmax, int64_t rows_per_group,
                        int64_t dest_group_rows, int32_t atomic, void* stream);
/* seen[i] += 1 where radii[i] > 0, i < n (the per-step "how many views saw this row" counts of log_amd.dist: what the
 * reference's step calls flag_vis, /root/reference/LoG/model/counter.py:48,50, summed over a rank's views). */
int lograst_add_visible(float* seen, const int32_t* radii, int64_t n, void* stream);
/* The same for k <= 16 views in one pass: `radii` is a HOST array of k device pointers ([n] int32 each); seen[i] += the number
 * of them with radii[j][i] > 0. */
int lograst_add_visible_n(float* seen, const int32_t* const* radii, int32_t k, int64_t n, void* stream);

/* ---- performance knobs -------------------------------------------------------------------------------------------
 * Launch-shape parameters that change no result (thresholds, grid caps, dispatch orders; the list is enumerated by
 * lograst_knob_count / lograst_knob_info).  Each is the environment variable of the same name unless overridden here;
 * overrides take effect from the next launch.  Sizing helpers (lograst_tile_state_bytes) and the forward of one view must
 * see the same values: change knobs between views, not between a sizing call and its forward.  log_amd.tune() calibrates
 * them per device.  All return 0 or LOGRAST_ERR_ARG (unknown name / value outside [lo, hi]). */
int lograst_knob_count(void);
int lograst_knob_info(int32_t index, const char** name, int32_t* dflt, int32_t* lo, int32_t* hi, const char** what);
int lograst_set_knob(const char* name, int32_t value);
int lograst_get_knob(const char* name, int32_t* value);
int lograst_reset_knobs(void);

/* Copies {num_instances, overflow_flag, longest tile list, rect instances} of a tile_state to host (synchronises
 * the stream; any pointer may be NULL).  rect_instances = what the plain rect rule of the reference would have
 * binned (num_rendered of the third-party package); num_instances <= rect_instances when the support cull is on. */
int lograst_read_state(const void* tile_state, uint32_t* num_instances_host, uint32_t* overflow_host,
                       uint32_t* max_tile_len_host, uint32_t* rect_instances_host, void* stream);

/* Support cull in the binning stage (default on; also LOGRAST_TILE_CULL=0): a tile of a Gaussian's rect becomes
 * a list entry only if the Gaussian can reach alpha >= 1/255 somewhere inside it.  Result-preserving (dropped
 * entries fail the alpha floor at every pixel of the tile); with it off the tile lists are exactly the
 * reference's rect lists.  Process-wide; returns the previous setting. */
int lograst_set_tile_cull(int enabled);

/* ---- backward (A6, A6b) ------------------------------------------------------------------------
 * Stands for _RasterizeGaussians.backward of the third-party package, triggered by loss.backward()
 * (LoG/utils/trainer.py:158).  dL_dimage[3,H,W] in; all gradient outputs are overwritten:
 *   dL_dmeans2d[n,3]  (x,y = d/d ndc, z = 0; consumed at LoG/model/counter.py:40,46)
 *   dL_dmeans3d[n,3], dL_dscales[n,3], dL_drotations[n,4], dL_dopacities[n], dL_dcolors[n,3]
 * bwd_rows[n, LOGRAST_BWD_ROW_FLOATS] is scratch: the reverse walk's accumulator rows (64-byte aligned; the forward's
 * bwd_scratch, or any zeroed block); the chain-rule kernel reads each live Gaussian's row and writes the separate
 * outputs (dL_dmeans2d for every row; dL_dopacities / dL_dcolors written, or added to when accumulating).  rotations /
 * dl_drotations must be 16-byte aligned (one 16-byte access per Gaussian).  flags:
 *   LOGRAST_BWD_SCRATCH_ZEROED  bwd_rows is already zeroed (the forward's bwd_scratch; or the caller's memset);
 *   LOGRAST_BWD_ACCUMULATE      multi-view accumulation (new, not in the reference): dl_dopacities, dl_dcolors,
 *                               dl_dmeans3d, dl_dscales, dl_drotations are running sums that this call ADDS to
 *                               (the reverse walk's atomics and the chain-rule kernel write straight into the
 *                               caller's per-step gradient bucket; no separate accumulate pass).
 *   LOGRAST_BWD_CONIC_TOUCHED_ONLY  bwd_rows is zeroed only in the rows of Gaussians with point_weight > 0 (what a
 *                               forward with extras and a bwd_scratch leaves on large inputs: see lograst_forward_render);
 *                               needs point_weight.  (Documentation of the caller's state: the chain rule skips the
 *                               rows with point_weight == 0 whenever point_weight is given, with or without this flag.)
 * point_weight (optional, NULL = none): the forward's per-Gaussian maximum blend weight.  A Gaussian with weight 0
 * contributed to no pixel, so its dL/dmean2D and dL/dconic are exactly zero: the chain rule skips it (its gradients
 * are written as 0, or left alone when accumulating) without reading its inputs -- in an opaque scene that is most of
 * the Gaussians.  It must be the UNMODIFIED point_weight output of the forward whose geom / tile_state / point_list are
 * passed here: the forward clears it for every row and only composited Gaussians (radii > 0) ever raise it, so the kernel
 * takes point_weight > 0 alone as the live flag and does not read radii for that decision.
 * Alignment: bwd_rows 64 bytes; rotations / dl_drotations 16 bytes; every other array 4 bytes (the kernels' full-width
 * clears of dl_dmeans2d / dl_dopacities / dl_dcolors / dl_dmeans3d / dl_dscales / dl_dcov3d start with a scalar head). */
#define LOGRAST_BWD_SCRATCH_ZEROED 1
#define LOGRAST_BWD_ACCUMULATE 2
#define LOGRAST_BWD_CONIC_TOUCHED_ONLY 4
/* multi-view accumulation into ONE 64-byte row per Gaussian (new): dl_dmeans3d points to the caller's running sums
 * [n][LOGRAST_GRAD_ROW_FLOATS] (64-byte aligned; slots 0-2 dL/dmeans3D, 3-5 dL/dscales, 6-9 dL/drotations, 10 dL/dopacity,
 * 11-13 dL/dcolour, 14-15 never touched) and every gradient of this view is ADDED there; dl_dscales, dl_drotations,
 * dl_dopacities, dl_dcolors are not used (NULL allowed).  A Gaussian that composited somewhere then costs the chain rule
 * one read-modify-write of one line instead of five pieces in five arrays (log_amd.dist.GradientBucket(row_major=True)).
 * Not with cov3d_precomp. */
#define LOGRAST_BWD_ACCUMULATE_ROWS 8
#define LOGRAST_GRAD_ROW_FLOATS 16

int lograst_backward(const lograst_view* view, int32_t n, const float* means3d, const float* scales,
                     const float* rotations, const int32_t* radii, const void* geom, const void* tile_state,
                     const uint32_t* point_list, const float* final_t, const int32_t* n_contrib,
                     const float* dl_dimage, float* dl_dmeans2d, float* bwd_rows, float* dl_dopacities,
                     float* dl_dcolors, float* dl_dmeans3d, float* dl_dscales, float* dl_drotations,
                     const float* point_weight, int32_t flags, void* stream);

/* Stage A6b alone: the per-Gaussian chain rule, given dL/d(ndc mean) [n,3] and dL/d(conic) [n,4] (as left
 * by the reverse walk).  Writes dl_dmeans3d/dl_dscales/dl_drotations.  lograst_backward = reverse walk +
 * this call; exposed so the two halves can be checked separately (the chain rule is ill-conditioned for
 * near-degenerate Gaussians, the walk is not). */
int lograst_project_backward(const lograst_view* view, int32_t n, const float* means3d, const float* scales,
                             const float* rotations, const int32_t* radii, const float* dl_dmeans2d,
                             const float* dl_dconic, float* dl_dmeans3d, float* dl_dscales,
                             float* dl_drotations, void* stream);

/* ---- "next" row N1: simple_knn._C.distCUDA2 ----------------------------------------------------------
 * Replaces distCUDA2 (third-party simple-knn, un-vendored; called at /root/reference/LoG/utils/file.py:88-91 and
 * LoG/model/base_gaussian.py:39-42): out[i] = mean squared distance from point i to its 3 nearest other
 * points (exact, fp32).  scratch: lograst_knn_scratch_bytes(p) bytes of device memory. */
size_t lograst_knn_scratch_bytes(int32_t p);
int lograst_knn_mean_dist2(int32_t p, const float* points, float* out, void* scratch, size_t scratch_bytes,
                           void* stream);

/* ---- "next" row N2: the packages' native `shs=` input --------------------------------------------------
 * colours[n,3] = max(0, 0.5 + sum_{k < (deg+1)^2} basis_k(normalise(mean - campos)) * shs[n,k,:]), shs laid out
 * [n, max_coeffs, 3]; clamped[n*3] (u8) records which channels hit the clamp.  Backward: dl_dshs[n,max_coeffs,3]
 * is overwritten (accumulate = 0) or added to (accumulate != 0: running sums over views), the direction gradient
 * is ADDED to dl_dmeans3d.  (Third-party package's computeColorFromSH;
 * LoG passes colors_precomp instead, LoG/render/renderer.py:144-145.) */
int lograst_sh_forward(int32_t n, int32_t degree, int32_t max_coeffs, const float* means3d, const float* campos,
                       const float* shs, float* colors, uint8_t* clamped, void* stream);
int lograst_sh_backward(int32_t n, int32_t degree, int32_t max_coeffs, const float* means3d, const float* campos,
                        const float* shs, const uint8_t* clamped, const float* dl_dcolors, float* dl_dshs,
                        float* dl_dmeans3d, int32_t accumulate, void* stream);

/* ---- debugging / test access to intermediates --------------------------------------------------
 * Pointers into a tile_state block (device): offsets has tiles+1 entries. */
const uint32_t* lograst_tile_offsets(const void* tile_state, int32_t width, int32_t height);
/* Lazily ordered tile lists (knob LOGRAST_LAZY_SORT, default on).  The third-party package sorts every (tile, depth) key
 * of a view (one global radix sort) before it composites; this library orders, of a list of more than 4096 keys, only
 * the first window (7680 positions, cut at a depth-bucket boundary) before the compositing pass, lets that pass mark the
 * tiles in which a pixel was still open at the end of the ordered part, and then orders the rest of exactly those lists
 * and composites their tiles again.  Images, fork maps, n_contrib, point_weight and every gradient are the same bit for
 * bit either way (tests/test_gpu_knobs.py); what differs is point_list behind the ordered part of a list nobody walked:
 * those entries are unspecified until lograst_finish_lists has run.
 *   lograst_ordered_lengths: lengths_out[t] (device, tiles entries) = leading positions of tile t's list that are in
 *     final order (the whole list for lists of up to 4096 keys, and for every list when the knob is 0);
 *   lograst_finish_lists: orders every list to its end (same tile_state / keys / point_list / capacity as the forward
 *     call, before any other forward reuses the keys buffer).  The lists are then what LOGRAST_LAZY_SORT=0 produces. */
int lograst_ordered_lengths(const void* tile_state, int32_t width, int32_t height, uint32_t* lengths_out, void* stream);
int lograst_finish_lists(void* tile_state, int32_t width, int32_t height, void* keys, uint32_t* point_list,
                         uint32_t capacity, void* stream);

/* ---- "next" row N3: level-of-detail selection ------------------------------------------------------------
 * Replaces TensorTree.traverse + TensorTree._query_tree_torch (/root/reference/LoG/model/tensor_tree.py:131-185;
 * caller LoG.prepare, LoG/model/level_of_gaussian.py:241) together with the Gaussian.compute_radius it calls per
 * level (level_of_gaussian.py:65-88: gather, exp / normalize activations, compute_radius_module.compute_radius).
 * Inputs are the tree buffers as TensorTree keeps them (node_index i32[num_points], -1 = leaf; tree
 * i32[num_nodes, max_child], -1 = removed child), the RAW model parameters (xyz[P,3], log-scales[P,3],
 * unnormalised quaternions[P,4]), the roots to start from (i64[num_roots]) and the camera of
 * lograst_compute_radius.  `levels` = number of levels to expand below the roots = min(tree.max_level, max_depth)
 * of the reference call (values beyond the tree's depth cost only empty launches).
 * out_index (i64, capacity out_capacity >= num_points is always enough: a point is selected at most once)
 * receives, in the reference's order, [roots kept | children kept at level 1 | ... | frontier left at the depth
 * limit], keep = (radius < min_resolution_pixel) | is_leaf.  The count stays on the device; read it with
 * lograst_lod_read (one stream synchronise -- the only one; the reference synchronises several times per level).
 * scratch: lograst_lod_scratch_bytes(num_roots, num_nodes, max_child) bytes. */
size_t lograst_lod_scratch_bytes(int32_t num_roots, int32_t num_nodes, int32_t max_child);
int lograst_lod_traverse(int32_t num_points, int32_t num_nodes, int32_t max_child, const int32_t* node_index,
                         const int32_t* tree, const float* xyz, const float* scaling, const float* rotation,
                         const int64_t* root_index, int32_t num_roots, const float* projmatrix,
                         const float* viewmatrix, float focal_x, float focal_y, float tanfovx, float tanfovy,
                         float min_resolution_pixel, int32_t levels, int64_t* out_index, uint32_t out_capacity,
                         void* scratch, size_t scratch_bytes, void* stream);
/* count_host / overflow_host / frontier_left_host: host words (the last two optional).  overflow != 0 means the
 * tree buffers were inconsistent (a point reachable twice) and out_index is incomplete.  frontier_left = how many of
 * the selected points are nodes that were still waiting to be expanded when `levels` ran out (0 whenever `levels`
 * reached the bottom of the tree): a caller that passes a cached tree depth as `levels` re-runs with the full value
 * if this is not 0. */
int lograst_lod_read(const void* scratch, uint32_t* count_host, uint32_t* overflow_host, uint32_t* frontier_left_host,
                     void* stream);

/* ---- "next" row N4: what LoG does with the rasterizer's outputs after every view ----------------------------
 * (a) lograst_id_histogram replaces `torch.unique(point_id_pixel, sorted=True, return_counts=True)` + dropping the
 *     leading -1 (/root/reference/LoG/render/renderer.py:156-159): ids_out (i32) = the distinct ids >= 0 in
 *     ascending order, counts_out (i64) = pixels each one wins; both need capacity min(n, num_pixels), n = number
 *     of Gaussians handed to the rasterizer (ids are < n).  The number of distinct ids stays on the device:
 *     lograst_id_histogram_read synchronises the stream and returns it.
 *     scratch: lograst_id_histogram_scratch_bytes(n). */
size_t lograst_id_histogram_scratch_bytes(int32_t n);
int lograst_id_histogram(int32_t n, const int32_t* point_id_pixel, int32_t num_pixels, int32_t* ids_out,
                         int64_t* counts_out, void* scratch, size_t scratch_bytes, void* stream);
int lograst_id_histogram_read(const void* scratch, uint32_t* count_host, void* stream);

/* (b) lograst_counter_update replaces Counter.update_by_output for ONE view
 *     (/root/reference/LoG/model/counter.py:36-68; caller LoG.update_by_output, level_of_gaussian.py:364-365).
 *     visible_index i64[nv] (no duplicates) maps the view's submitted Gaussians to model rows; grad_means2d
 *     f32[nv,3] is viewspace_points.grad; radii i32[nv]; point_weight f32[nv]; point_id i32[k] / point_count i64[k]
 *     the lists of (a).  The eight Counter buffers (length num_points, dtypes as registered at counter.py:7-19)
 *     are updated in place.  flag_vis_out (u8[nv], optional) receives radii > 0 (counter.py:48,50). */
int lograst_counter_update(int32_t nv, const int64_t* visible_index, const float* grad_means2d, const int32_t* radii,
                           const float* point_weight, int32_t k, const int32_t* point_id, const int64_t* point_count,
                           int32_t num_points, float* weights_max, float* weights_sum, float* grad_sum,
                           int16_t* radii_max, int16_t* visible_count, int32_t* radii_max_max, int32_t* area_sum,
                           int32_t* create_steps, uint8_t* flag_vis_out, void* stream);

/* (c) lograst_sparse_adam replaces SparseOptimizer.step + _single_tensor_adam
 *     (/root/reference/LoG/model/sparse_optimizer.py:41-78,163-196; caller LoG.step, level_of_gaussian.py:379-390):
 *     for every r < m with flag_vis[r] != 0 and every key, Adam on model row index[r] starting from the gathered
 *     parameter row r, with the reference's scalars: step_size = lr / (1 - beta1^step), bias_correction2_sqrt =
 *     sqrt(1 - beta2^step), eps; amsgrad when max_exp_avg_sq != NULL.  All keys go in one launch (num_keys <= 8). */
typedef struct lograst_adam_key {
  void* model_param;        /* f32 [num_points, width]: rows index[flag_vis] are rewritten */
  const void* param;        /* f32 [m, width]: params[key].data */
  const void* grad;         /* f32 [m, width]: params[key].grad */
  void* exp_avg;            /* f32 [num_points, width] */
  void* exp_avg_sq;         /* f32 [num_points, width] */
  void* max_exp_avg_sq;     /* f32 [num_points, width] or NULL */
  int32_t width;            /* floats per row */
  float step_size;          /* lr / bias_correction1 */
} lograst_adam_key;
int lograst_sparse_adam(int32_t m, int32_t num_points, const int64_t* index, const uint8_t* flag_vis,
                        int32_t num_keys, const lograst_adam_key* keys, double beta1, double beta2,
                        double bias_correction2_sqrt, double eps, void* stream);

/* Version 4: lograst_activate_backward (below) and lograst_sparse_adam in ONE launch, for steps of a single view (the
 * reference's trainer: one backward, then SparseOptimizer.step).  For every r < n with radii[r] > 0 (LoG's flag_vis,
 * LoG/model/counter.py:48) the raw gradients of row r are computed as lograst_activate_backward computes them (dl_dact_xyz
 * passes through) and applied at once to model row index[r] and its moments as lograst_sparse_adam applies them -- the
 * gradients are never written.  keys[6] in the order xyz, scaling, opacity, rotation, colors, shs (widths 3, 3, 1, 4, 3,
 * 3 * sh_coeffs; `grad` is ignored, `param` = the gathered raw rows [n, width]; model_param == NULL: key not optimised).
 * Same op sequences as the two kernels: the model and the moments come out bit for bit the same. */
int lograst_activate_backward_adam(int32_t n, const float* raw_xyz, const float* raw_scaling, const float* raw_opacity,
                                   const float* raw_rotation, int32_t sh_coeffs, int32_t active_degree,
                                   const float* camera_center, const float* dl_dact_xyz, const float* dl_dact_scaling,
                                   const float* dl_dact_opacity, const float* dl_dact_rotation, const float* dl_dact_colors,
                                   int32_t num_points, const int64_t* index, const int32_t* radii,
                                   const lograst_adam_key* keys, double beta1, double beta2, double bias_correction2_sqrt,
                                   double eps, void* stream);

/* ---- rows N2 / N3: LoG.get_all + Activation.activate_root_return, fused ---------------------------------------
 * Replaces the per-key gathers of LoG.get_all (/root/reference/LoG/model/level_of_gaussian.py:262-296) and the
 * activations of Activation.activate_root_return / colors_activation (/root/reference/LoG/model/activation.py:27-44;
 * SH polynomial /root/reference/LoG/model/sh_utils.py:31-72).  For every r < n, row index[r] of the model buffers
 * (xyz[P,3], scaling[P,3] log-scales, opacity[P,1] logits, rotation[P,4], colors[P,3] SH DC term, shs[P,K,3] higher
 * coefficients, K = 0 allowed with shs = NULL) is copied to the raw_* outputs [n, ...] (the step's parameters) and
 * activated: act_scaling = exp, act_opacity = sigmoid, act_rotation = q / max(|q|, 1e-12), act_colors =
 * 0.28209479 * colors + 0.5 (+ eval_sh_wobase(normalize(xyz - camera_center), shs, active_degree) when
 * active_degree > 0; degrees 1..3, camera_center = 3 floats on the device).  xyz is passed through (raw_xyz). */
int lograst_gather_activate(int32_t n, int32_t num_points, const int64_t* index, const float* xyz,
                            const float* scaling, const float* opacity, const float* rotation, const float* colors,
                            const float* shs, int32_t sh_coeffs, int32_t active_degree, const float* camera_center,
                            float* raw_xyz, float* raw_scaling, float* raw_opacity, float* raw_rotation,
                            float* raw_colors, float* raw_shs, float* act_scaling, float* act_opacity,
                            float* act_rotation, float* act_colors, void* stream);
/* Backward of the activations for the first n rows (the rows that are parameters): from dL/d(act_*) to
 * dL/d(raw_*).  dl_dshs (may be NULL) is [n, sh_coeffs, 3]; coefficients above active_degree get 0.  The direction
 * is detached in the reference (activation.py:30), so xyz receives no gradient from the colours. */
int lograst_activate_backward(int32_t n, const float* raw_xyz, const float* raw_scaling, const float* raw_opacity,
                              const float* raw_rotation, int32_t sh_coeffs, int32_t active_degree,
                              const float* camera_center, const float* dl_dact_scaling, const float* dl_dact_opacity,
                              const float* dl_dact_rotation, const float* dl_dact_colors, float* dl_dscaling,
                              float* dl_dopacity, float* dl_drotation, float* dl_dcolors, float* dl_dshs, void* stream);

/* ---- per-kernel timing with HIP events on the launch stream (used by bench.py) -----------------
 * When enabled every kernel launch is bracketed by hipEventRecord on its stream.  read() synchronises
 * the recorded events and returns, for kernel slot i < LOGRAST_NUM_KERNELS, accumulated milliseconds
 * and launch counts since the last reset. */
#define LOGRAST_NUM_KERNELS 20
void lograst_profile_enable(int on);
void lograst_profile_reset(void);
int lograst_profile_read(double* ms_out, int64_t* count_out);
const char* lograst_kernel_name(int slot);

#ifdef __cplusplus
}
#endif
#endif /* LOGRAST_H */

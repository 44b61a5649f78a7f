"""Host-side mirror of the rasterizer packages LoG imports.

Reproduces, name for name, the Python surface of ``diff_gaussian_rasterization`` (graphdeco upstream)
and ``diff_gaussian_rasterization_wodilate`` (chingswy fork, branch ``antialias``) as LoG uses them
(/root/reference/LoG/render/renderer.py:1,57-78,100-107,141-165,190-198;
/root/reference/LoG/model/level_of_gaussian.py:59,73-78,207-221):

  * ``GaussianRasterizationSettings`` -- 12-field NamedTuple, keyword constructed (renderer.py:63-76);
  * ``GaussianRasterizer(raster_settings=...)`` -- nn.Module with ``.raster_settings``, ``__call__`` with the
    keyword set of renderer.py:141-153 (+ ``use_filter`` for the fork), ``compute_radius`` (fork,
    level_of_gaussian.py:59) and ``markVisible``;
  * autograd: differentiable w.r.t. means3D, means2D (NDC-scaled screen-space gradient), colors_precomp,
    opacities, scales, rotations;
  * returns ``(image, radii)`` for the upstream flavour and
    ``(image, radii, point_id_pixel, point_weight_pixel, point_weight)`` for the fork (renderer.py:154-165).

All arithmetic runs in hand-written HIP kernels (log_amd/csrc) through the C ABI of liblograst.so
(include/lograst.h).  Tensors must live on the MI355X; there is no CPU fallback.
"""
import ctypes
import threading
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class Flavour(NamedTuple):
    """What distinguishes the two third-party packages on this path."""
    name: str
    filter_mode: int   # low-pass when use_filter is on
    ndc_cull: int      # |ndc| > 1.3 cull (LoG/cuda/compute_radius_kernel.cu:131-134)
    extras: int        # 5-tuple return


UPSTREAM = Flavour("diff_gaussian_rasterization", _lib.FILTER_DILATE, 0, 0)
WODILATE = Flavour("diff_gaussian_rasterization_wodilate", _lib.FILTER_CLAMP, 1, 1)

# ---- instance-capacity policy ----------------------------------------------------------------------
# Default (None): exact -- stage 1 reports the number of tile instances to the host (one 4-byte
# read-back, as the third-party package does for `num_rendered`) and the buffers are sized exactly.
# With a hint, no host synchronisation happens in forward() (one C-ABI call per forward); the kernels refuse to
# write past the capacity, or to sort a list longer than the max_tile_len hint, and record the overflow in the
# process's status block (include/lograst.h: LOGRAST_STATUS_*), which `overflow_since_reset()` / bench.py read.
_capacity_hint = None
_max_len_hint = 0
# Default mode (no hint): SPECULATIVE stage 2 (lograst_forward_speculative) -- the forward is enqueued in one piece with
# buffers sized from a running estimate (instances per Gaussian of the recent forwards at this resolution x 1.25) while
# the exact count travels to the host on a side stream; in the rare case the estimate was too small the kernels of the
# speculative stage 2 returned without rendering and stage 2 is repeated with exact buffers.  Always exact results, one
# read-back per forward like the third-party package's `num_rendered`, but the stream never waits for the host.
# set_speculative(False) restores the two-call form (stage 1, read-back, exact allocation, stage 2).
# Reverse walk: views whose Gaussians average fewer tile instances than this are walked in the row-split form (four 4x4
# blocks per wave), the others one 8x8 quadrant per wave (measured on the MI355X: 30 M random Gaussians, 1.46 instances
# each: 949 -> 700 us; C2, 2.8: 292 -> 307; a tree-ordered heavy-tailed selection, 2.3: 448 -> 579).
ROWSPLIT_MAX_INSTANCES_PER_GAUSSIAN = 1.8
_speculative = True
# In-place leaf gradients are OPT-IN (round-3 advisory): a custom Function cannot see how the engine was invoked, so
# adding into .grad behind autograd's back breaks torch.autograd.grad(loss, leaves), backward(inputs=[subset]) and
# AccumulateGrad hooks (DDP / FSDP reducers).  Off globally; a caller that owns the leaves opts them in --
# log_amd.dist.GradientBucket.attach() does (their .grad ARE views of its bucket), or set_inplace_leaf_grads(True).
_inplace_leaf_grads = False
_INPLACE_TAG = "_lograst_inplace_grad"   # attribute set on leaves whose owner asked for the in-place route
_status = {}          # device -> int32[8] status block (sticky across forwards, all streams)
_debug_keep = False   # tests: keep dL/dconic of the last backward (HipBackend.last_conic_grad)
_hit_masks = True     # a training forward hands its compositing kernels a hit-mask buffer for the reverse walk (blend.hip)
_zero_hit_masks = False   # tools/mask_stats.py: a zero-filled buffer, so that the slots nobody wrote read as "no visit"
_keep_keys = False    # tests: the forward's key buffer stays alive in `saved` (finish_lists below needs it)
_DEBUG_ADDR = bool(int(__import__('os').environ.get('LOGRAST_DEBUG_ADDR', '0')))


def set_instance_capacity(n, max_tile_len=0):
    """n = int: sync-free forward with room for n tile instances; None: exact sizing (default).
    max_tile_len: upper bound on the longest per-tile list in sync-free mode (0 = unknown: the sort then launches
    every multi-block level the capacity allows; only lists longer than 8192 entries care).  A forward whose real
    numbers exceed either renders nothing and raises the overflow bit of the status block."""
    global _capacity_hint, _max_len_hint
    _capacity_hint = None if n is None else int(n)
    _max_len_hint = int(max_tile_len) if n is not None else 0


def set_hit_masks(enabled):
    """Training forwards allocate lograst_view.hit_masks (128 B per walked (tile, 64-entry chunk)) so that the reverse walk
    reuses the forward's support decisions instead of recomputing them (default on; speed only).  -> previous value."""
    global _hit_masks
    prev, _hit_masks = _hit_masks, bool(enabled)
    return prev


def set_speculative(enabled):
    """Default-mode forward: True (default) = speculative stage 2 from the running capacity estimate, False = stage 1,
    read-back, exact allocation, stage 2.  Returns the previous setting."""
    global _speculative
    prev, _speculative = _speculative, bool(enabled)
    return prev


def set_inplace_leaf_grads(enabled):
    """Opt-in (default False).  True: when every differentiable input of a rasterizer call is a leaf that already has a
    dense fp32 ``.grad`` (a multi-view step after its first view, or grads that are views of a flat bucket), backward adds
    into those tensors in place (LOGRAST_BWD_ACCUMULATE) instead of returning fresh gradients for autograd to add (five
    full passes over the attributes per view).  Same sums for a plain ``loss.backward()``; NOT compatible with
    ``torch.autograd.grad``, ``backward(inputs=...)`` or AccumulateGrad-node hooks (DDP / FSDP): those never see the
    gradient.  Leaves with tensor hooks always take the autograd route.  Individual leaves are opted in by
    ``allow_inplace_grad(tensor)`` (what ``GradientBucket.attach`` does).  Returns the previous setting."""
    global _inplace_leaf_grads
    prev, _inplace_leaf_grads = _inplace_leaf_grads, bool(enabled)
    return prev


def allow_inplace_grad(t, enabled=True):
    """Mark one leaf: a backward whose five differentiable inputs are ALL marked (or set_inplace_leaf_grads(True)) adds
    into their existing .grad in place.  The caller asserts that it only ever runs plain ``.backward()`` on them."""
    setattr(t, _INPLACE_TAG, bool(enabled))
    return t


class _CapacityModel:
    """Running estimate of the tile-instance count per (device, W, H, band): the largest instances-per-Gaussian ratio,
    instance count and tile-list length of the recent forwards, slowly forgotten so that one dense view does not pin its
    buffers for the rest of the run.  Only a guess is needed: a wrong one costs a repeated stage 2, never a result."""
    HEADROOM, LEN_HEADROOM, DECAY, FIRST_RATIO = 1.25, 1.5, 0.98, 2.0

    def __init__(self):
        self.hist = {}
        self.retries = 0      # speculative attempts that had to be repeated (diagnostics)
        self.forwards = 0
        self.lock = threading.Lock()   # render threads share the model (round-3 advisory)

    def ratio(self, key):
        with self.lock:
            h = self.hist.get(key)
            return h["ratio"] if h else 0.0

    def guess(self, key, n, tiles):
        with self.lock:
            h = self.hist.get(key)
            if h is None:
                return int(self.FIRST_RATIO * n) + 4 * tiles + 4096, 0
            cap = int(max(h["I"], h["ratio"] * n) * self.HEADROOM) + 4096
            return min(cap, 0x7fffffff), int(h["L"] * self.LEN_HEADROOM) + 256

    def update(self, key, n, instances, max_len, retried):
        with self.lock:
            self.forwards += 1
            self.retries += int(retried)
            h = self.hist.get(key)
            ratio = instances / max(n, 1)
            if h is None:
                self.hist[key] = dict(I=float(instances), ratio=ratio, L=float(max_len))
            else:
                h["I"] = max(float(instances), h["I"] * self.DECAY)
                h["ratio"] = max(ratio, h["ratio"] * self.DECAY)
                h["L"] = max(float(max_len), h["L"] * self.DECAY)

    def reset(self):
        with self.lock:
            self.hist.clear()
            self.retries = self.forwards = 0


_cap_model = _CapacityModel()


def capacity_stats(reset=False):
    """dict(forwards, retries) of the speculative default mode since the last reset (no synchronisation)."""
    out = dict(forwards=_cap_model.forwards, retries=_cap_model.retries)
    if reset:
        _cap_model.reset()
    return out


def _status_block(device):
    st = _status.get(device)
    if st is None:
        st = torch.zeros(8, dtype=torch.int32, device=device)
        _status[device] = st
    return st


def _current_device():
    return torch.device("cuda", torch.cuda.current_device())


def last_state_info(device=None):
    """(num_instances, overflowed, longest_tile_list, rect_instances) of the most recent forward on `device`
    (synchronises).  rect_instances = instances of the reference's plain rect rule (>= num_instances when the
    support cull is on)."""
    st = _status.get(device if device is not None else _current_device())
    if st is None:
        return 0, False, 0, 0
    w = st.cpu().tolist()
    return int(w[1]), bool(w[2]), int(w[3]), int(w[4])


def last_overflow(device=None):
    """(num_instances, overflowed) of the most recent forward (synchronises)."""
    n, o = last_state_info(device)[:2]
    return n, o


def overflow_since_reset(device=None, reset=True):
    """dict(overflowed, max_instances, max_tile_len, forwards) over EVERY forward on `device` (all streams) since the
    last reset -- what a caller of the sync-free mode checks after a batch of views (synchronises)."""
    st = _status.get(device if device is not None else _current_device())
    if st is None:
        return dict(overflowed=False, max_instances=0, max_tile_len=0, forwards=0)
    w = st.cpu().tolist()
    if reset:
        st.zero_()
    return dict(overflowed=bool(w[0]), max_instances=int(w[5]), max_tile_len=int(w[6]), forwards=int(w[7]))


def set_tile_cull(enabled):
    """Support cull in the binning stage (default on): tiles of a Gaussian's rect in which it cannot reach
    alpha >= 1/255 are not binned.  Result-preserving; off = the reference's exact rect lists.  Returns the
    previous setting."""
    return bool(_lib.lib().lograst_set_tile_cull(1 if enabled else 0))


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(0)


def _dev_f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


# Blocks carved out of one allocation are read / written in lock step by one kernel; starting them at multiples of
# n floats apart puts all streams on the same HBM channels (measured: the activation backward 0.12 -> 0.17 ms for
# n = 1 M).  Each block is followed by this many floats (4352 B: keeps 16-byte alignment, breaks the stride).
_BLOCK_SKEW = 1088
_ITEMSIZE = {torch.float32: 4, torch.int32: 4, torch.uint8: 1, torch.int64: 8}


class _TileRows:
    """The band of tile rows rendered by the calls inside a ``tile_rows(begin, end)`` block (image split across GPUs:
    log_amd/dist.py).  (0, 0) = the whole image."""

    def __init__(self):
        self._local = threading.local()

    def get(self):
        return getattr(self._local, "rows", (0, 0))

    def set(self, rows):
        self._local.rows = rows


_tile_rows = _TileRows()


# ---- pinning the compositing kernels' form (round-4 verdict, weak #9) ------------------------------------------------
# Both forms give bit-identical forwards and gradients within the same tolerance; which one is FASTER depends on the view
# (tile instances per Gaussian).  By default the package decides per call from what it knows -- the resolution's recent
# history for the forward, the forward's own count for the backward -- so the time a host sees depends on what rendered
# before.  A host that wants reproducible timings (or knows its scene) pins the form: per rasterizer object
# (``GaussianRasterizer(raster_settings=..., walk_form="rows")``), for a block of calls (``with walk_form("quadrant"):``)
# or process-wide (``set_walk_form``); the backward of a view uses what its forward was pinned to.
_FORMS = {None: 0, "auto": 0, "rows": 1, "quadrant": 2}       # include/lograst.h: LOGRAST_FORM_*
_walk_form_default = 0
_walk_form_local = threading.local()


def _form_code(form):
    if form not in _FORMS:
        raise ValueError("walk_form must be None / 'auto', 'rows' or 'quadrant' (got %r)" % (form,))
    return _FORMS[form]


def set_walk_form(form):
    """Process-wide pin of the compositing kernels' form: 'rows' (four 4x4 blocks per wave: tiny splats), 'quadrant' (one
    8x8 quadrant per wave), or None / 'auto' (default: decided per call).  Returns the previous setting's name."""
    global _walk_form_default
    prev, _walk_form_default = _walk_form_default, _form_code(form)
    return {0: "auto", 1: "rows", 2: "quadrant"}[prev]


def _pinned_form():
    return getattr(_walk_form_local, "form", 0) or _walk_form_default


class walk_form:
    """``with walk_form("rows"): ...`` pins the form for the rasterizer calls of this thread inside the block."""

    def __init__(self, form):
        self.form = _form_code(form)

    def __enter__(self):
        self.prev = getattr(_walk_form_local, "form", 0)
        _walk_form_local.form = self.form
        return self

    def __exit__(self, *exc):
        _walk_form_local.form = self.prev
        return False


class tile_rows:
    """``with tile_rows(begin, end): image, ... = rasterizer(...); loss.backward()`` renders (and differentiates) only
    the tile rows [begin, end) -- pixel rows [16*begin, 16*end) -- of every view set up inside the block; the backward
    of a view uses the band its forward used.  New design, not part of the reference's API."""

    def __init__(self, begin, end):
        self.rows = (int(begin), int(end))

    def __enter__(self):
        self.prev = _tile_rows.get()
        _tile_rows.set(self.rows)
        return self

    def __exit__(self, *exc):
        _tile_rows.set(self.prev)
        return False


class HipBackend:
    """Drives the kernels.  All buffers come from torch's caching allocator."""

    @staticmethod
    def require(device):
        if device.type != "cuda":
            raise _lib.LograstError(
                f"log_amd rasterizer needs tensors on the MI355X (got device '{device}'); "
                "the HIP kernels are the only implementation -- there is no CPU fallback")
        return _lib.lib()

    def make_view(self, rs, flavour, use_filter, device, cov3D=None, g_cov3D=None):
        """cov3D / g_cov3D: the `cov3D_precomp` input ([N, 6]) and, for a backward, where dL/dcov3D goes."""
        keep = (_dev_f32(rs.viewmatrix, device), _dev_f32(rs.projmatrix, device), _dev_f32(rs.bg, device).reshape(-1))
        if keep[0].numel() != 16 or keep[1].numel() != 16 or keep[2].numel() != 3:
            raise ValueError("viewmatrix/projmatrix must be 4x4 and bg must have 3 entries")
        v = _lib.LograstView()
        v.width, v.height = int(rs.image_width), int(rs.image_height)
        v.tanfovx, v.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
        v.scale_modifier = float(rs.scale_modifier)
        v.filter_mode = flavour.filter_mode if use_filter else _lib.FILTER_NONE
        v.ndc_cull, v.extras = flavour.ndc_cull, flavour.extras
        v.viewmatrix, v.projmatrix, v.bg = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
        v.tile_row_begin, v.tile_row_end = _tile_rows.get()
        v.cov3d_precomp = cov3D.data_ptr() if cov3D is not None else None
        v.dl_dcov3d = g_cov3D.data_ptr() if g_cov3D is not None else None
        v.walk_form = _lib.FORM_AUTO
        return v, keep

    @staticmethod
    def walk_form(instances, n):
        """LOGRAST_FORM_* for a view of n Gaussians with `instances` tile instances (None / 0 = unknown)."""
        if not instances or n <= 0:
            return _lib.FORM_AUTO
        return _lib.FORM_ROWS if instances < ROWSPLIT_MAX_INSTANCES_PER_GAUSSIAN * n else _lib.FORM_QUADRANT

    last_forms = None   # {"fwd": "rows" | "quadrant", "bwd": ...} of the most recent forward / backward (diagnostics)

    def _note_form(self, kind, walk_form, n):
        """Which compositing kernel the library launches for this call: the rule of lr_launch_blend_fwd / _bwd
        (log_amd/csrc/blend.hip) restated -- knob LOGRAST_{FWD,BWD}_ROWS, else the view's hint, else (reverse walk only)
        row-split from LOGRAST_HELPER_MIN_N Gaussians.  bench.py / the parity tests record it next to their numbers."""
        L = _lib.lib()
        val = ctypes.c_int32(2)
        L.lograst_get_knob(("LOGRAST_%s_ROWS" % kind.upper()).encode(), ctypes.byref(val))
        if val.value != 2:
            rows = val.value == 1
        elif walk_form != _lib.FORM_AUTO:
            rows = walk_form == _lib.FORM_ROWS
        elif kind == "bwd":
            L.lograst_get_knob(b"LOGRAST_HELPER_MIN_N", ctypes.byref(val))
            rows = n >= val.value
        else:
            rows = False
        if self.last_forms is None:
            self.last_forms = {}
        self.last_forms[kind] = "rows" if rows else "quadrant"

    @staticmethod
    def _carve(device, parts):
        """One allocation for several buffers: parts = [(name, dtype, shape)], every buffer 256-byte aligned inside it
        (one caching-allocator call instead of one per buffer: the host side of a forward is mostly such calls)."""
        spans, total = [], 0
        for _, dt, shape in parts:
            n = _ITEMSIZE[dt]
            for d in shape:
                n *= int(d)
            spans.append((total, n))
            total += (n + 255) // 256 * 256
        arena = torch.empty(max(total, 1), dtype=torch.uint8, device=device)
        return {name: arena[off:off + n].view(dt).view(shape) for (name, dt, shape), (off, n) in zip(parts, spans)}

    def forward(self, rs, flavour, use_filter, means3D, scales, rotations, opacities, colors, scratch_floats=0,
                cov3D=None):
        """scratch_floats: non-zero = allocate the backward's accumulator rows (16 fp32 = 64 B per Gaussian,
        include/lograst.h: LOGRAST_BWD_ROW_FLOATS) and have the forward clear them: saved as `bwd_scratch` and consumed by
        the first backward.
        cov3D: the rasterizer's cov3D_precomp ([N, 6] fp32) instead of scales / rotations (both None then)."""
        device = means3D.device
        L = self.require(device)
        N = means3D.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        view, keep = self.make_view(rs, flavour, use_filter, device, cov3D=cov3D)
        stream = _stream_ptr(device)
        status = _status_block(device)
        i32, f32, u8 = torch.int32, torch.float32, torch.uint8
        # what the caller gets (image, radii, the fork's maps) / what backward needs / what dies with this call
        # (every output is an allocation of its own, like the third-party packages': views of one arena would share a
        # version counter -- an in-place op on `radii` would invalidate `image` for autograd -- and any one of them kept
        # alive would pin all the others' memory)
        o = {"image": torch.empty(3, H, W, dtype=f32, device=device), "radii": torch.empty(N, dtype=i32, device=device)}
        if flavour.extras:
            o.update(pid=torch.empty(H, W, dtype=i32, device=device), pwp=torch.empty(H, W, dtype=f32, device=device),
                     pw=torch.empty(N, dtype=f32, device=device))
        kept = [("geom", u8, (L.lograst_geom_bytes(N),)), ("state", u8, (L.lograst_tile_state_bytes(W, H, N),)),
                ("final_T", f32, (H, W)), ("n_contrib", i32, (H, W))]
        scratch_floats = _lib.BWD_ROW_FLOATS if scratch_floats else 0
        if scratch_floats and N:
            kept.append(("bwd_scratch", f32, (N * scratch_floats,)))
        instances = None
        want_masks = bool(_hit_masks and scratch_floats and N)
        if want_masks:   # (knob LOGRAST_HIT_MASKS = 0: the library would ignore the buffer, and a backward under a different
            kv = ctypes.c_int32(1)   # knob value must not find an unwritten one)
            L.lograst_get_knob(b"LOGRAST_HIT_MASKS", ctypes.byref(kv))
            want_masks = kv.value != 0

        def masks_for(cap):
            """The hit-mask buffer of a training forward with room for `cap` tile instances (uninitialised)."""
            if not want_masks:
                return None
            m = (torch.zeros if _zero_hit_masks else torch.empty)(L.lograst_hit_mask_bytes(cap, W, H) // 8, dtype=torch.int64,
                                                                  device=device)
            view.hit_masks, view.hit_mask_words = m.data_ptr(), m.numel()
            return m
        ckey = (device.index, W, H, _tile_rows.get())
        hist_ratio = _cap_model.ratio(ckey)
        # which form the compositing kernel takes: instances per Gaussian as the recent forwards of this resolution had
        # them (speculative / exact mode), or the caller's capacity (sync-free mode)
        pin = _pinned_form()
        view.walk_form = pin or self.walk_form(_capacity_hint if _capacity_hint is not None else hist_ratio * N, N)
        self._note_form("fwd", view.walk_form, N)
        with torch.cuda.device(device):
            if _capacity_hint is None and _speculative and N > 0:
                tiles = ((W + 15) // 16) * ((H + 15) // 16)
                capacity, max_len = _cap_model.guess(ckey, N, tiles)
                # (the guessed list is an allocation of its own: inside the arena it would stay pinned, at its guessed
                # size and next to the exact one after a retry, until backward -- round-3 advisory)
                k = self._carve(device, kept)
                plist = torch.empty(capacity, dtype=i32, device=device)
                keys = torch.empty(L.lograst_keys_bytes(capacity), dtype=u8, device=device)
                masks = masks_for(capacity)
                n_host, m_host = ctypes.c_uint32(0), ctypes.c_uint32(0)
                _lib.check(L.lograst_forward_speculative(
                    ctypes.byref(view), N, _ptr(means3D), _ptr(scales), _ptr(rotations), _ptr(opacities), _ptr(colors),
                    _ptr(o["radii"]), _ptr(k["geom"]), _ptr(k["state"]), _ptr(keys), _ptr(plist), capacity, max_len,
                    _ptr(o["image"]), _ptr(k["final_T"]), _ptr(k["n_contrib"]), _ptr(o.get("pid")), _ptr(o.get("pwp")),
                    _ptr(o.get("pw")), _ptr(k.get("bwd_scratch")), scratch_floats if "bwd_scratch" in k else 0,
                    _ptr(status), ctypes.byref(n_host), ctypes.byref(m_host), stream))
                n_inst, n_len = int(n_host.value), int(m_host.value)
                instances = n_inst
                retry = n_inst > capacity or (max_len != 0 and n_len > max_len)
                _cap_model.update(ckey, N, n_inst, n_len, retry)
                if retry:   # the speculative stage 2 rendered nothing: once more with exact buffers
                    capacity, max_len = n_inst, max(n_len, 1)
                    plist = torch.empty(capacity, dtype=i32, device=device)
                    keys = torch.empty(L.lograst_keys_bytes(capacity), dtype=u8, device=device)
                    masks = masks_for(capacity)
                    _lib.check(L.lograst_forward_render(
                        ctypes.byref(view), N, _ptr(k["geom"]), _ptr(k["state"]), _ptr(keys), _ptr(plist), capacity, max_len,
                        _ptr(o["image"]), _ptr(k["final_T"]), _ptr(k["n_contrib"]), _ptr(o.get("pid")), _ptr(o.get("pwp")),
                        _ptr(o.get("pw")), _ptr(k.get("bwd_scratch")), scratch_floats if "bwd_scratch" in k else 0,
                        _ptr(status), stream))
            elif _capacity_hint is None:
                k = self._carve(device, kept)
                n_host, m_host = ctypes.c_uint32(0), ctypes.c_uint32(0)
                _lib.check(L.lograst_forward_project(ctypes.byref(view), N, _ptr(means3D), _ptr(scales), _ptr(rotations),
                                                     _ptr(opacities), _ptr(colors), _ptr(o["radii"]), _ptr(k["geom"]),
                                                     _ptr(k["state"]), ctypes.byref(n_host), ctypes.byref(m_host), stream))
                capacity, max_len = int(n_host.value), max(int(m_host.value), 1)
                plist = torch.empty(capacity, dtype=i32, device=device)
                keys = torch.empty(L.lograst_keys_bytes(capacity), dtype=u8, device=device)
                masks = masks_for(capacity)
                _lib.check(L.lograst_forward_render(
                    ctypes.byref(view), N, _ptr(k["geom"]), _ptr(k["state"]), _ptr(keys), _ptr(plist), capacity, max_len,
                    _ptr(o["image"]), _ptr(k["final_T"]), _ptr(k["n_contrib"]), _ptr(o.get("pid")), _ptr(o.get("pwp")),
                    _ptr(o.get("pw")), _ptr(k.get("bwd_scratch")), scratch_floats if "bwd_scratch" in k else 0,
                    _ptr(status), stream))
            else:
                capacity, max_len = _capacity_hint, _max_len_hint
                k = self._carve(device, kept + [("plist", i32, (capacity,))])
                plist = k["plist"]
                keys = torch.empty(L.lograst_keys_bytes(capacity), dtype=u8, device=device)
                masks = masks_for(capacity)
                _lib.check(L.lograst_forward(
                    ctypes.byref(view), N, _ptr(means3D), _ptr(scales), _ptr(rotations), _ptr(opacities), _ptr(colors),
                    _ptr(o["radii"]), _ptr(k["geom"]), _ptr(k["state"]), _ptr(keys), _ptr(plist), capacity, max_len,
                    _ptr(o["image"]), _ptr(k["final_T"]), _ptr(k["n_contrib"]), _ptr(o.get("pid")), _ptr(o.get("pwp")),
                    _ptr(o.get("pw")), _ptr(k.get("bwd_scratch")), scratch_floats if "bwd_scratch" in k else 0,
                    _ptr(status), stream))
        if instances is None:
            instances = capacity          # exact mode: the real count; sync-free: the caller's (tight) upper bound
        if _DEBUG_ADDR:
            print("fwd state@%x keys@%x capacity=%d stream=%x" % (k["state"].data_ptr(), keys.data_ptr(), capacity, stream.value or 0), flush=True)
        kept_keys = (keys, capacity) if _keep_keys else None
        del keys, keep
        saved = dict(radii=o["radii"], geom=k["geom"].view(f32), state=k["state"].view(i32), plist=plist,
                     final_T=k["final_T"], n_contrib=k["n_contrib"], bwd_scratch=k.get("bwd_scratch"),
                     point_weight=o.get("pw"), tile_rows=(view.tile_row_begin, view.tile_row_end), instances=int(instances),
                     walk_form_pin=pin, hit_masks=masks if want_masks else None,
                     hit_mask_form=int(L.lograst_forward_form(ctypes.byref(view))) if want_masks else 0)
        if kept_keys is not None:
            saved["keys"], saved["capacity"] = kept_keys
        return o["image"], o["radii"], o.get("pid"), o.get("pwp"), o.get("pw"), saved

    def backward(self, rs, flavour, use_filter, means3D, scales, rotations, saved, grad_image, sink=None, cov3D=None):
        """sink: optional dict of running-sum tensors (means3D, scales, rotations, opacities, colors) that this call
        adds into (LOGRAST_BWD_ACCUMULATE); the corresponding returned gradients are then None.
        cov3D (the forward's cov3D_precomp; no sink): the last two results are (dL/dcov3D [N, 6], None)."""
        device = means3D.device
        L = self.require(device)
        N = means3D.shape[0]
        g_cov = None
        if cov3D is not None:
            if sink is not None:
                raise ValueError("a gradient sink has no cov3D_precomp entry")
            g_cov = torch.empty(N, 6, dtype=torch.float32, device=device)
        view, keep = self.make_view(rs, flavour, use_filter, device, cov3D=cov3D, g_cov3D=g_cov)
        view.tile_row_begin, view.tile_row_end = saved.get("tile_rows", (0, 0))   # the band the forward rendered
        f32 = dict(dtype=torch.float32, device=device)
        grad_image = grad_image.to(torch.float32).contiguous()
        need = _lib.BWD_ROW_FLOATS
        # The reverse walk adds into ONE 64-byte accumulator row per Gaussian (mean x y, conic A B C, opacity, colour):
        # the block the forward already cleared for this purpose (first backward of this forward), else a fresh
        # torch.zeros.  In the 5-tuple flavour the forward cleared the rows of the contributing Gaussians only
        # (point_weight > 0); the chain rule skips all the others (LOGRAST_BWD_CONIC_TOUCHED_ONLY) -- and hands out the
        # separate outputs: dL/dmeans2D, and dL/dopacity / dL/dcolour (written, or added into the sink's running sums).
        acc = saved.pop("bwd_scratch", None)
        pw = saved.get("point_weight")
        flags = 1
        if acc is None or acc.numel() < need * N:
            acc = torch.zeros(N * need, **f32)
        elif pw is not None:
            flags |= 4
        # tiny splats -> the row-split reverse walk; a form pinned for the forward holds for its backward
        view.walk_form = saved.get("walk_form_pin", 0) or self.walk_form(saved.get("instances", 0), N)
        self._note_form("bwd", view.walk_form, N)
        hm = saved.get("hit_masks")      # the forward's support ballots (lograst_view.hit_masks): the reverse walk reads them
        if hm is not None:
            view.hit_masks, view.hit_mask_words = hm.data_ptr(), hm.numel()
            view.hit_mask_form = max(int(saved.get("hit_mask_form", 0)), 0)
        g_conic = acc          # (the C ABI's `bwd_rows`)
        g_means2D = torch.empty(N, 3, **f32)
        if sink is None:
            g = self._carve(device, [("rot", torch.float32, (N, 4)), ("means3D", torch.float32, (N, 3)),
                                     ("scales", torch.float32, (N, 3)), ("colors", torch.float32, (N, 3)),
                                     ("opac", torch.float32, (N,))])
            g_means3D, g_scales, g_rot, g_colors, g_opac = g["means3D"], g["scales"], g["rot"], g["colors"], g["opac"]
        elif "rows" in sink:   # one 64-byte row of running sums per Gaussian (LOGRAST_BWD_ACCUMULATE_ROWS)
            g_means3D, g_scales, g_rot, g_opac, g_colors = sink["rows"], None, None, None, None
            flags |= 8
        else:
            g_opac, g_colors = sink["opacities"], sink["colors"]
            g_means3D, g_scales, g_rot = sink["means3D"], sink["scales"], sink["rotations"]
            flags |= 2
        with torch.cuda.device(device):
            _lib.check(L.lograst_backward(ctypes.byref(view), N, _ptr(means3D), _ptr(scales), _ptr(rotations),
                                          _ptr(saved["radii"]), _ptr(saved["geom"]), _ptr(saved["state"]),
                                          _ptr(saved["plist"]), _ptr(saved["final_T"]), _ptr(saved["n_contrib"]),
                                          _ptr(grad_image), _ptr(g_means2D), _ptr(g_conic), _ptr(g_opac),
                                          _ptr(g_colors), _ptr(g_means3D), _ptr(g_scales), _ptr(g_rot), _ptr(pw),
                                          flags, _stream_ptr(device)))
        del keep
        # test introspection only: dL/dconic [N, 4] (A, B, C, 0) out of the accumulator rows
        self.last_conic_grad = (torch.cat([acc.view(N, need)[:, 2:5], acc.new_zeros(N, 1)], dim=1)
                                if _debug_keep and N else (acc.new_zeros(0, 4) if _debug_keep else None))
        if sink is not None:
            return None, g_means2D, None, None, None, None
        if cov3D is not None:
            return g_means3D, g_means2D, g_colors, g_opac, g_cov, None
        return g_means3D, g_means2D, g_colors, g_opac, g_scales, g_rot

    def sh_forward(self, means3D, campos, shs, degree):
        """Native `shs=` path (lograst_sh_forward): colours[N,3] and the clamp mask u8[N,3]."""
        device = means3D.device
        L = self.require(device)
        N, M = shs.shape[0], shs.shape[1]
        cp = _dev_f32(campos, device).reshape(-1)
        colors = torch.empty(N, 3, dtype=torch.float32, device=device)
        clamped = torch.empty(N, 3, dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            _lib.check(L.lograst_sh_forward(N, int(degree), M, _ptr(means3D), _ptr(cp), _ptr(shs), _ptr(colors),
                                            _ptr(clamped), _stream_ptr(device)))
        return colors, clamped

    def sh_backward(self, means3D, campos, shs, degree, clamped, g_colors, g_means3D, into=None):
        """dL/dshs: a new tensor, or added into `into` (running sum; returns None).  The view-direction gradient is
        added into g_means3D in place."""
        device = means3D.device
        L = self.require(device)
        N, M = shs.shape[0], shs.shape[1]
        cp = _dev_f32(campos, device).reshape(-1)
        g_shs = torch.empty(N, M, 3, dtype=torch.float32, device=device) if into is None else into
        with torch.cuda.device(device):
            _lib.check(L.lograst_sh_backward(N, int(degree), M, _ptr(means3D), _ptr(cp), _ptr(shs), _ptr(clamped),
                                             _ptr(g_colors), _ptr(g_shs), _ptr(g_means3D), 0 if into is None else 1,
                                             _stream_ptr(device)))
        return g_shs if into is None else None

    def project_backward(self, rs, flavour, use_filter, means3D, scales, rotations, radii, g_means2D, g_conic,
                         cov3D=None):
        """Stage A6b alone (lograst_project_backward): used by the parity tests.  With cov3D: -> (dL/dmeans3D,
        dL/dcov3D, None)."""
        device = means3D.device
        L = self.require(device)
        N = means3D.shape[0]
        g_cov = torch.empty(N, 6, dtype=torch.float32, device=device) if cov3D is not None else None
        view, keep = self.make_view(rs, flavour, use_filter, device, cov3D=cov3D, g_cov3D=g_cov)
        f32 = dict(dtype=torch.float32, device=device)
        g_means3D, g_scales, g_rot = torch.empty(N, 3, **f32), torch.empty(N, 3, **f32), torch.empty(N, 4, **f32)
        with torch.cuda.device(device):
            _lib.check(L.lograst_project_backward(ctypes.byref(view), N, _ptr(means3D), _ptr(scales), _ptr(rotations),
                                                  _ptr(radii), _ptr(g_means2D), _ptr(g_conic), _ptr(g_means3D),
                                                  _ptr(g_scales), _ptr(g_rot), _stream_ptr(device)))
        del keep
        if cov3D is not None:
            return g_means3D, g_cov, None
        return g_means3D, g_scales, g_rot

    def tile_rows(self, rs, flavour, use_filter, means3D, scales, rotations):
        """-> (y0, y1) int32[N]: the tile rows [y0, y1) each Gaussian's rect covers on the whole image (y0 = y1 = 0 when
        the projection drops it): lograst_tile_rows.  See log_amd.dist.band_index."""
        device = means3D.device
        L = self.require(device)
        N = means3D.shape[0]
        m, s, r = _dev_f32(means3D, device), _dev_f32(scales, device), _dev_f32(rotations, device)
        view, keep = self.make_view(rs, flavour, use_filter, device)
        rows = torch.empty(N, dtype=torch.int32, device=device)
        with torch.cuda.device(device):
            _lib.check(L.lograst_tile_rows(ctypes.byref(view), N, _ptr(m), _ptr(s), _ptr(r), _ptr(rows), _stream_ptr(device)))
        del keep
        return rows & 0xffff, (rows >> 16) & 0xffff

    def compute_radius(self, means3D, scales, rotations, projmatrix, viewmatrix, fx, fy, tanfovx, tanfovy):
        device = means3D.device
        L = self.require(device)
        P = means3D.shape[0]
        m, s, r = _dev_f32(means3D, device), _dev_f32(scales, device), _dev_f32(rotations, device)
        pm, vm = _dev_f32(projmatrix, device), _dev_f32(viewmatrix, device)
        out = torch.empty(P, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            _lib.check(L.lograst_compute_radius(P, _ptr(m), _ptr(s), _ptr(r), _ptr(pm), _ptr(vm), float(fx), float(fy),
                                                float(tanfovx), float(tanfovy), _ptr(out), _stream_ptr(device)))
        return out


    def lod_traverse(self, node_index, tree, xyz, scaling, rotation, root_index, projmatrix, viewmatrix, fx, fy,
                     tanfovx, tanfovy, min_resolution_pixel, levels, depth_hint=None):
        """N3 (log_amd/lod.py): -> int64 indices selected for this camera, in the reference's order.  depth_hint: a
        (possibly stale) guess of the tree's depth; fewer levels are launched, and the call repeats itself with the
        full `levels` if the device reports that the guess cut the descent short."""
        device = xyz.device
        L = self.require(device)
        P = int(xyz.shape[0])
        ni = node_index.detach().to(device=device, dtype=torch.int32).contiguous()
        tr = tree.detach().to(device=device, dtype=torch.int32).contiguous()
        num_nodes, max_child = (int(tr.shape[0]), int(tr.shape[1])) if tr.dim() == 2 else (0, 1)
        roots = root_index.detach().to(device=device, dtype=torch.int64).contiguous()
        x, s, r = _dev_f32(xyz, device), _dev_f32(scaling, device), _dev_f32(rotation, device)
        pm, vm = _dev_f32(projmatrix, device), _dev_f32(viewmatrix, device)
        out = torch.empty(max(P, 1), dtype=torch.int64, device=device)
        nbytes = L.lograst_lod_scratch_bytes(int(roots.numel()), num_nodes, max_child)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=device)
        count, overflow, left = ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_uint32(0)
        tries = [int(levels)] if depth_hint is None or depth_hint >= levels else [int(depth_hint), int(levels)]
        with torch.cuda.device(device):
            for lv in tries:
                _lib.check(L.lograst_lod_traverse(P, num_nodes, max_child, _ptr(ni), _ptr(tr), _ptr(x), _ptr(s), _ptr(r),
                                                  _ptr(roots), int(roots.numel()), _ptr(pm), _ptr(vm), float(fx),
                                                  float(fy), float(tanfovx), float(tanfovy), float(min_resolution_pixel),
                                                  lv, _ptr(out), int(out.numel()), _ptr(scratch), nbytes,
                                                  _stream_ptr(device)))
                _lib.check(L.lograst_lod_read(_ptr(scratch), ctypes.byref(count), ctypes.byref(overflow),
                                              ctypes.byref(left), _stream_ptr(device)))
                if left.value == 0:
                    break
        if overflow.value:
            raise _lib.LograstError("lod_traverse: inconsistent tree buffers (a point is reachable more than once)")
        return out[:count.value]


    def id_histogram(self, point_id_pixel, n):
        """N4a (log_amd/counter.py): sorted distinct ids >= 0 of the per-pixel id map and their pixel counts."""
        device = point_id_pixel.device
        L = self.require(device)
        pid = point_id_pixel.detach().to(torch.int32).contiguous()
        npix = int(pid.numel())
        cap = max(1, min(n, npix))
        ids = torch.empty(cap, dtype=torch.int32, device=device)
        counts = torch.empty(cap, dtype=torch.int64, device=device)
        nbytes = L.lograst_id_histogram_scratch_bytes(n)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=device)
        k = ctypes.c_uint32(0)
        with torch.cuda.device(device):
            _lib.check(L.lograst_id_histogram(n, _ptr(pid), npix, _ptr(ids), _ptr(counts), _ptr(scratch), nbytes,
                                              _stream_ptr(device)))
            _lib.check(L.lograst_id_histogram_read(_ptr(scratch), ctypes.byref(k), _stream_ptr(device)))
        return ids[:k.value], counts[:k.value]

    def counter_update(self, buffers, visible_index, grad, radii, point_weight, point_id, point_count):
        """N4b (log_amd/counter.py): Counter.update_by_output for one view; -> flag_vis (bool[nv])."""
        device = radii.device
        L = self.require(device)
        want = {"weights_max": torch.float32, "weights_sum": torch.float32, "grad_sum": torch.float32,
                "radii_max": torch.int16, "visible_count": torch.int16, "radii_max_max": torch.int32,
                "area_sum": torch.int32, "create_steps": torch.int32}
        for name, dt in want.items():
            b = buffers[name]
            if b.dtype != dt or b.device != device or not b.is_contiguous():
                raise ValueError(f"counter buffer {name}: expected a contiguous {dt} tensor on {device}")
        num_points = int(buffers["weights_max"].shape[0])
        vi = visible_index.detach().to(device=device, dtype=torch.int64).contiguous()
        nv = int(vi.numel())
        g = _dev_f32(grad, device)
        if g.shape != (nv, 3):
            raise ValueError("viewspace gradient must be [nv, 3]")
        r = radii.detach().to(torch.int32).contiguous()
        w = _dev_f32(point_weight, device).reshape(-1)
        if int(r.numel()) != nv or int(w.numel()) != nv:
            raise ValueError("radii and point_weight must have one entry per visible_index row")
        pid = point_id.detach().to(device=device, dtype=torch.int32).contiguous()
        pc = point_count.detach().to(device=device, dtype=torch.int64).contiguous()
        flag = torch.empty(nv, dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            _lib.check(L.lograst_counter_update(
                nv, _ptr(vi), _ptr(g), _ptr(r), _ptr(w), int(pid.numel()), _ptr(pid), _ptr(pc), num_points,
                *[_ptr(buffers[name]) for name in want], _ptr(flag), _stream_ptr(device)))
        return flag.view(torch.bool)

    def sparse_adam(self, index, flag_vis, entries, beta1, beta2, bias_correction2_sqrt, eps):
        """N4c (log_amd/sparse_optimizer.py).  entries: (model_param, param, grad, exp_avg, exp_avg_sq,
        max_exp_avg_sq | None, step_size) per key."""
        device = index.device
        L = self.require(device)
        idx = index.detach().to(torch.int64).contiguous()
        m = int(idx.numel())
        fv = flag_vis.detach().to(device=device).contiguous()
        fv = fv.view(torch.uint8) if fv.dtype == torch.bool else fv.to(torch.uint8)
        if int(fv.numel()) != m:
            raise ValueError("flag_vis and index must have the same length")
        keep = []
        with torch.cuda.device(device):
            for first in range(0, len(entries), 8):
                chunk = entries[first:first + 8]
                keys = (_lib.LograstAdamKey * len(chunk))()
                for slot, (model_p, param, grad, m1, m2, mmax, step_size) in zip(keys, chunk):
                    num_points = int(model_p.shape[0])
                    width = int(model_p[0].numel()) if num_points else 1
                    for t in (model_p, m1, m2) + ((mmax,) if mmax is not None else ()):
                        if t.device != device or t.dtype != torch.float32 or not t.is_contiguous() or t.shape != model_p.shape:
                            raise ValueError("parameters and Adam moments must be contiguous fp32 tensors on the parameter's "
                                             "device (log_amd.sparse_optimizer.step moves host-resident moments there)")
                    p, g = _dev_f32(param, device), _dev_f32(grad, device)
                    if int(p.numel()) != m * width or int(g.numel()) != m * width:
                        raise ValueError("param / grad rows do not match index")
                    keep += [p, g]
                    slot.model_param, slot.param, slot.grad = model_p.data_ptr(), p.data_ptr(), g.data_ptr()
                    slot.exp_avg, slot.exp_avg_sq = m1.data_ptr(), m2.data_ptr()
                    slot.max_exp_avg_sq = mmax.data_ptr() if mmax is not None else None
                    slot.width, slot.step_size = width, float(step_size)
                _lib.check(L.lograst_sparse_adam(m, num_points, _ptr(idx), _ptr(fv), len(chunk), keys, float(beta1),
                                                 float(beta2), float(bias_correction2_sqrt), float(eps),
                                                 _stream_ptr(device)))
        del keep


    def gather_activate(self, index, bufs, degree, campos):
        """Rows N2/N3 (log_amd/get_all.py): gather rows `index` of the model buffers, -> (raw dict, activated dict)."""
        device = bufs["xyz"].device
        L = self.require(device)
        idx = index.detach().to(device=device, dtype=torch.int64).contiguous()
        n, P = int(idx.numel()), int(bufs["xyz"].shape[0])
        src = {k: _dev_f32(v, device) for k, v in bufs.items()}
        K = int(src["shs"].shape[1]) if "shs" in src else 0
        # one allocation for all outputs (the quaternion blocks first: their rows are read and written as float4)
        widths = [("raw", "rotation", 4), ("act", "rotation", 4), ("raw", "xyz", 3), ("raw", "scaling", 3),
                  ("raw", "colors", 3), ("act", "scaling", 3), ("act", "colors", 3), ("raw", "opacity", 1),
                  ("act", "opacity", 1)] + ([("raw", "shs", 3 * K)] if K else [])
        flat = torch.empty(sum(n * w + _BLOCK_SKEW for _, _, w in widths), dtype=torch.float32, device=device)
        raw, act, off = {}, {}, 0
        for kind, key, w in widths:
            view = flat[off:off + n * w].view((n, K, 3) if key == "shs" else (n, w))
            (raw if kind == "raw" else act)[key] = view
            off += n * w + _BLOCK_SKEW
        cp = _dev_f32(campos, device).reshape(-1) if campos is not None else None
        with torch.cuda.device(device):
            _lib.check(L.lograst_gather_activate(
                n, P, _ptr(idx), _ptr(src["xyz"]), _ptr(src["scaling"]), _ptr(src["opacity"]), _ptr(src["rotation"]),
                _ptr(src["colors"]), _ptr(src.get("shs")), K, int(degree), _ptr(cp), _ptr(raw["xyz"]),
                _ptr(raw["scaling"]), _ptr(raw["opacity"]), _ptr(raw["rotation"]), _ptr(raw["colors"]),
                _ptr(raw.get("shs")), _ptr(act["scaling"]), _ptr(act["opacity"]), _ptr(act["rotation"]),
                _ptr(act["colors"]), _stream_ptr(device)))
        act["xyz"] = raw["xyz"]
        return raw, act

    def activate_backward_adam(self, raw, n, degree, campos, g_xyz, g_scaling, g_opacity, g_rotation, g_colors, index, radii,
                               entries, beta1, beta2, bias_correction2_sqrt, eps):
        """Activation backward + sparse Adam in one launch (lograst_activate_backward_adam; log_amd.get_all's fused step).
        entries: {key: (model_param, exp_avg, exp_avg_sq, max_exp_avg_sq | None, step_size)} for the keys that are optimised
        (of xyz / scaling / opacity / rotation / colors / shs)."""
        device = raw["xyz"].device
        L = self.require(device)
        K = int(raw["shs"].shape[1]) if "shs" in raw else 0
        ups = [_dev_f32(t, device) for t in (g_xyz, g_scaling, g_opacity, g_rotation, g_colors)]
        cp = _dev_f32(campos, device).reshape(-1) if campos is not None else None
        idx = index.detach().to(torch.int64).contiguous()
        rad = radii.detach().to(torch.int32).contiguous()
        if int(idx.numel()) < n or int(rad.numel()) < n:
            raise ValueError("index / radii are shorter than the rows that are parameters")
        keys = (_lib.LograstAdamKey * 6)()
        num_points = int(next(iter(entries.values()))[0].shape[0])
        for slot, key in zip(keys, ("xyz", "scaling", "opacity", "rotation", "colors", "shs")):
            if key not in entries:
                continue
            model_p, m1, m2, mmax, step_size = entries[key]
            width = int(model_p[0].numel()) if num_points else 1
            for t in (model_p, m1, m2) + ((mmax,) if mmax is not None else ()):
                if t.device != device or t.dtype != torch.float32 or not t.is_contiguous() or t.shape != model_p.shape:
                    raise ValueError("parameters and Adam moments must be contiguous fp32 tensors on the parameter's device")
            p = raw[key]
            if not p.is_contiguous() or int(p.numel()) < n * width:
                raise ValueError("gathered parameter rows do not match")
            slot.model_param, slot.param, slot.grad = model_p.data_ptr(), p.data_ptr(), None
            slot.exp_avg, slot.exp_avg_sq = m1.data_ptr(), m2.data_ptr()
            slot.max_exp_avg_sq = mmax.data_ptr() if mmax is not None else None
            slot.width, slot.step_size = width, float(step_size)
        with torch.cuda.device(device):
            _lib.check(L.lograst_activate_backward_adam(
                int(n), _ptr(raw["xyz"]), _ptr(raw["scaling"]), _ptr(raw["opacity"]), _ptr(raw["rotation"]), K, int(degree),
                _ptr(cp), _ptr(ups[0]), _ptr(ups[1]), _ptr(ups[2]), _ptr(ups[3]), _ptr(ups[4]), num_points, _ptr(idx),
                _ptr(rad), keys, float(beta1), float(beta2), float(bias_correction2_sqrt), float(eps), _stream_ptr(device)))
        del ups

    def activate_backward(self, raw, n, degree, campos, g_scaling, g_opacity, g_rotation, g_colors):
        """-> dict of dL/d(raw rows [0, n)) for scaling / opacity / rotation / colors (/ shs when degree > 0)."""
        device = raw["xyz"].device
        L = self.require(device)
        K = int(raw["shs"].shape[1]) if "shs" in raw else 0
        # one allocation; 16-byte aligned blocks first (quaternion rows and, for 3K % 4 == 0, the SH rows go out as float4)
        widths = [("rotation", 4)] + ([("shs", 3 * K)] if K and degree > 0 else []) + [("scaling", 3), ("colors", 3), ("opacity", 1)]
        flat = torch.empty(sum(n * w + _BLOCK_SKEW for _, w in widths), dtype=torch.float32, device=device)
        g, off = {}, 0
        for key, w in widths:
            g[key] = flat[off:off + n * w].view((n, K, 3) if key == "shs" else (n, w))
            off += n * w + _BLOCK_SKEW
        ups = [_dev_f32(t, device) for t in (g_scaling, g_opacity, g_rotation, g_colors)]
        cp = _dev_f32(campos, device).reshape(-1) if campos is not None else None
        with torch.cuda.device(device):
            _lib.check(L.lograst_activate_backward(
                int(n), _ptr(raw["xyz"]), _ptr(raw["scaling"]), _ptr(raw["opacity"]), _ptr(raw["rotation"]), K,
                int(degree), _ptr(cp), _ptr(ups[0]), _ptr(ups[1]), _ptr(ups[2]), _ptr(ups[3]), _ptr(g["scaling"]),
                _ptr(g["opacity"]), _ptr(g["rotation"]), _ptr(g["colors"]), _ptr(g.get("shs")), _stream_ptr(device)))
        del ups
        return g


_backend = HipBackend()
_backward_view = threading.local()     # .radii: the radii of the forward whose backward ran last on this thread (see backward())


def last_backward_radii():
    """The `radii` output of the rasterizer forward whose backward node ran last on this thread, or None."""
    return getattr(_backward_view, "radii", None)


# ---- multi-view gradient accumulation (new design; not part of the reference's API) ---------------
_grad_sink = None


class accumulate_grads_into:
    """Context manager.  While active, every rasterizer backward ADDS its gradients w.r.t. means3D / scales /
    rotations / opacities / colors_precomp (or, with an "shs" entry, the SH coefficients) straight into the given fp32 tensors (e.g. the views of a
    log_amd.dist.GradientBucket) instead of returning them to autograd: the reverse walk's atomics and the
    chain-rule kernel write into the step's running sums, so a multi-view step needs no per-view accumulate pass.
    means2D (per-view, consumed by LoG's Counter) is still returned normally.  Row-major form: ``{"rows": [N, 16]}`` -- one
    64-byte row of running sums per Gaussian (columns 0-2 means3D, 3-5 scales, 6-9 rotations, 10 opacity, 11-13 colour;
    include/lograst.h: LOGRAST_BWD_ACCUMULATE_ROWS), what log_amd.dist.GradientBucket(row_major=True).sink() hands out.
    Otherwise the sink tensors must be
    contiguous fp32 [N,3],[N,3],[N,4],[N,1] or [N],[N,3] on the inputs' device; inputs routed through the sink get
    no autograd gradient."""

    def __init__(self, sink):
        if "rows" in sink:   # row-major running sums (log_amd.dist.GradientBucket(row_major=True).sink())
            t = sink["rows"]
            if (t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 2 or t.shape[1] != _lib.GRAD_ROW_FLOATS
                    or t.data_ptr() % 64):
                raise ValueError("gradient sink 'rows' must be a contiguous, 64-byte aligned float32 [N, 16] tensor")
            if "shs" in sink:
                raise ValueError("the row-major gradient sink has no native-SH form (pass colors_precomp)")
            self.sink = {"rows": t}
            return
        need = ("means3D", "scales", "rotations", "opacities") + (() if "shs" in sink else ("colors",))
        missing = [k for k in need if k not in sink]
        if missing:
            raise KeyError(f"gradient sink lacks {missing}")
        for k in need + (("shs",) if "shs" in sink else ()):
            t = sink[k]
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError(f"gradient sink '{k}' must be a contiguous float32 tensor")
        self.sink = dict(sink)

    def __enter__(self):
        global _grad_sink
        self.prev, _grad_sink = _grad_sink, self.sink
        return self

    def __exit__(self, *exc):
        global _grad_sink
        _grad_sink = self.prev
        return False


def _leaf_grad_sink(leaves, device):
    """The inputs' own ``.grad`` tensors as a gradient sink, or None unless EVERY differentiable input qualifies: a leaf
    that requires grad, without tensor hooks, whose .grad already exists as a dense contiguous fp32 tensor of its shape on
    this device (then adding in place is exactly what autograd's AccumulateGrad would do with a returned gradient)."""
    if torch.is_grad_enabled() or device.type != "cuda":   # create_graph=True: leave everything to autograd
        return None
    out = {}
    for name, t in zip(("means3D", "colors", "opacities", "scales", "rotations"), leaves):
        if t is None or not t.requires_grad or not t.is_leaf or t._backward_hooks:
            return None
        if getattr(t, "_post_accumulate_grad_hooks", None):
            return None
        g = t.grad
        if (g is None or g.dtype != torch.float32 or g.device != device or g.layout != torch.strided or
                not g.is_contiguous() or g.shape != t.shape or g.requires_grad):
            return None
        out[name] = g
    return out


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, colors, shs, opacities, scales, rotations, rs, flavour, use_filter, cov3D=None):
        m = means3D.detach().to(torch.float32).contiguous()
        cov = None
        if cov3D is not None:   # the packages' cov3D_precomp input (not LoG's path): scales / rotations are absent
            cov = cov3D.detach().to(torch.float32).contiguous()
            if cov.shape != (m.shape[0], 6):
                raise ValueError("cov3D_precomp must be [N, 6]")
            if _grad_sink is not None:
                raise ValueError("accumulate_grads_into has no entry for cov3D_precomp")
            s, r = m.new_empty(0, 3), m.new_empty(0, 4)    # placeholders for save_for_backward: never read
        else:
            s = scales.detach().to(torch.float32).contiguous()
            r = rotations.detach().to(torch.float32).contiguous()
        o = opacities.detach().to(torch.float32).contiguous().reshape(-1)
        n = m.shape[0]
        sh = clamped = None
        if shs is not None:   # the packages' native SH input (not LoG's path)
            sh = shs.detach().to(torch.float32).contiguous()
            if sh.dim() != 3 or sh.shape[0] != n or sh.shape[2] != 3 or sh.shape[1] < (int(rs.sh_degree) + 1) ** 2:
                raise ValueError("shs must be [N, >=(sh_degree+1)^2, 3]")
            c, clamped = _backend.sh_forward(m, rs.campos, sh, int(rs.sh_degree))
        else:
            c = colors.detach().to(torch.float32).contiguous()
        if not ((cov is not None or (s.shape == (n, 3) and r.shape == (n, 4))) and c.shape == (n, 3) and o.shape[0] == n
                and m.shape == (n, 3)):
            raise ValueError("rasterizer inputs must be means3D[N,3], scales[N,3], rotations[N,4], "
                             "colors_precomp[N,3], opacities[N,1]")
        wants_grad = any(ctx.needs_input_grad[:7])   # all False under torch.no_grad()
        wants_grad = wants_grad or (cov is not None and ctx.needs_input_grad[10])
        scratch_floats = _lib.BWD_ROW_FLOATS if wants_grad else 0
        if cov is not None:
            image, radii, pid, pwp, pw, saved = _backend.forward(rs, flavour, use_filter, m, None, None, o, c,
                                                                 scratch_floats=scratch_floats, cov3D=cov)
        else:
            image, radii, pid, pwp, pw, saved = _backend.forward(rs, flavour, use_filter, m, s, r, o, c,
                                                                 scratch_floats=scratch_floats)
        ctx.cov = cov
        ctx.rs, ctx.flavour, ctx.use_filter = rs, flavour, use_filter
        ctx.set_materialize_grads(False)   # no zero-filled gradients for radii / the fork maps (4 fill kernels per view)
        ctx.saved = saved
        # the backward trusts the forward's point_weight (which Gaussians it may skip, which dL/dconic rows are cleared):
        # an in-place change of that output between forward and backward is refused, like autograd does for saved tensors
        ctx.pw_version = pw._version if pw is not None else None
        ctx.leaves = (means3D, colors, opacities, scales, rotations) if (sh is None and cov is None) else None
        ctx.sh = (sh, clamped)
        ctx.shapes = (means2D.shape, opacities.shape)
        ctx.save_for_backward(m, s, r)
        if flavour.extras:
            ctx.mark_non_differentiable(radii, pid, pwp, pw)
            return image, radii, pid, pwp, pw
        ctx.mark_non_differentiable(radii)
        return image, radii

    @staticmethod
    def backward(ctx, grad_image, *unused):
        m, s, r = ctx.saved_tensors
        if grad_image is None:   # only non-differentiable outputs were used downstream
            return (None,) * 11
        # which view's backward is running: nodes further down the same graph (log_amd.get_all's fused step) read the
        # visibility of THIS render from here (the rasterizer's node runs before the nodes that produced its inputs)
        _backward_view.radii = ctx.saved.get("radii") if isinstance(ctx.saved, dict) else None
        m2_shape, o_shape = ctx.shapes
        sh, clamped = ctx.sh
        pw = ctx.saved.get("point_weight") if isinstance(ctx.saved, dict) else None
        if pw is not None and ctx.pw_version is not None and pw._version != ctx.pw_version:
            raise RuntimeError("the rasterizer's point_weight output was modified in place between forward and backward; "
                               "the backward uses it to skip Gaussians that contributed to no pixel -- clone it first")
        if ctx.cov is not None:
            g_m3, g_m2, g_c, g_o, g_cov, _ = _backend.backward(ctx.rs, ctx.flavour, ctx.use_filter, m, None, None,
                                                               ctx.saved, grad_image, cov3D=ctx.cov)
            g_sh = None
            if sh is not None:
                g_sh = _backend.sh_backward(m, ctx.rs.campos, sh, int(ctx.rs.sh_degree), clamped, g_c.contiguous(), g_m3)
                g_c = None
            return g_m3, g_m2.reshape(m2_shape), g_c, g_sh, g_o.reshape(o_shape), None, None, None, None, None, g_cov
        sink = _grad_sink
        if sink is None and ctx.leaves is not None and (
                _inplace_leaf_grads or all(getattr(t, _INPLACE_TAG, False) for t in ctx.leaves if t is not None)):
            sink = _leaf_grad_sink(ctx.leaves, m.device)
        if sink is not None and "rows" in sink:
            n = m.shape[0]
            if sh is not None:
                raise ValueError("the row-major gradient sink has no native-SH form (pass colors_precomp)")
            if sink["rows"].shape[0] != n or sink["rows"].device != m.device:
                raise ValueError("gradient sink does not match the rasterizer inputs")
            _, g_m2, _, _, _, _ = _backend.backward(ctx.rs, ctx.flavour, ctx.use_filter, m, s, r, ctx.saved, grad_image,
                                                    sink=sink)
            return None, g_m2.reshape(m2_shape), None, None, None, None, None, None, None, None, None
        if sink is not None and (sh is None or "shs" in sink):
            n = m.shape[0]
            if not (sink["means3D"].shape == (n, 3) and sink["scales"].shape == (n, 3) and
                    sink["rotations"].shape == (n, 4) and sink["opacities"].numel() == n and
                    (sh is not None or sink["colors"].shape == (n, 3)) and sink["means3D"].device == m.device):
                raise ValueError("gradient sink does not match the rasterizer inputs")
            if sh is None:
                _, g_m2, _, _, _, _ = _backend.backward(ctx.rs, ctx.flavour, ctx.use_filter, m, s, r, ctx.saved,
                                                        grad_image, sink=sink)
            else:
                # colours are an intermediate here: their gradient goes to a zeroed scratch, then through the SH
                # polynomial into the running dL/dshs (and the direction term into the running dL/dmeans3D)
                if sink["shs"].shape != sh.shape or not sink["shs"].is_contiguous():
                    raise ValueError("gradient sink 'shs' must be a contiguous tensor shaped like shs")
                g_c = torch.zeros(n, 3, dtype=torch.float32, device=m.device)
                _, g_m2, _, _, _, _ = _backend.backward(ctx.rs, ctx.flavour, ctx.use_filter, m, s, r, ctx.saved,
                                                        grad_image, sink=dict(sink, colors=g_c))
                _backend.sh_backward(m, ctx.rs.campos, sh, int(ctx.rs.sh_degree), clamped, g_c, sink["means3D"],
                                     into=sink["shs"])
            return None, g_m2.reshape(m2_shape), None, None, None, None, None, None, None, None, None
        g_m3, g_m2, g_c, g_o, g_s, g_r = _backend.backward(ctx.rs, ctx.flavour, ctx.use_filter, m, s, r, ctx.saved,
                                                           grad_image)
        g_sh = None
        if sh is not None:
            g_sh = _backend.sh_backward(m, ctx.rs.campos, sh, int(ctx.rs.sh_degree), clamped, g_c.contiguous(), g_m3)
            g_c = None
        return g_m3, g_m2.reshape(m2_shape), g_c, g_sh, g_o.reshape(o_shape), g_s, g_r, None, None, None, None


class GaussianRasterizer(nn.Module):
    FLAVOUR = WODILATE

    def __init__(self, raster_settings, walk_form=None):
        """walk_form (extension; the third-party packages take raster_settings only): pin the compositing kernels' form
        for this object's calls -- 'rows' / 'quadrant' / None (decided per call); see ``log_amd.rasterizer.walk_form``."""
        super().__init__()
        self.raster_settings = raster_settings
        self.walk_form = walk_form
        _form_code(walk_form)

    def markVisible(self, positions):
        """Frustum test of the third-party package (not called by LoG): view z > 0.2."""
        with torch.no_grad():
            vm = self.raster_settings.viewmatrix.to(positions.device, torch.float32)
            z = positions.to(torch.float32) @ vm[:3, 2] + vm[3, 2]
            return z > 0.2

    def compute_radius(self, xyz, scaling, rotation):
        """Fork-only method (level_of_gaussian.py:59): projected radius per point, 0 = not visible."""
        rs = self.raster_settings
        fx = rs.image_width / (2.0 * rs.tanfovx)
        fy = rs.image_height / (2.0 * rs.tanfovy)
        with torch.no_grad():
            return _backend.compute_radius(xyz, scaling * rs.scale_modifier, rotation, rs.projmatrix, rs.viewmatrix,
                                           fx, fy, rs.tanfovx, rs.tanfovy)

    def tile_rows(self, xyz, scaling, rotation, use_filter=True):
        """New (image split across GPUs): (y0, y1) int32[N], the tile rows [y0, y1) of each Gaussian's rect on
        the whole image under this rasterizer's settings, (0, 0) when the forward would drop it.  The Gaussians a forward
        inside ``tile_rows(b, e)`` keeps are exactly those with y0 < e and y1 > b (log_amd.dist.band_index)."""
        with torch.no_grad():
            return _backend.tile_rows(self.raster_settings, self.FLAVOUR, use_filter, xyz, scaling, rotation)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, **kwargs):
        flavour = self.FLAVOUR
        use_filter = True
        if "use_filter" in kwargs:
            if not flavour.extras:
                raise TypeError("forward() got an unexpected keyword argument 'use_filter'")
            use_filter = bool(kwargs.pop("use_filter"))
        if kwargs:
            raise TypeError(f"forward() got unexpected keyword arguments {sorted(kwargs)}")
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if shs is not None and not 0 <= int(self.raster_settings.sh_degree) <= 3:
            raise ValueError("sh_degree must be 0..3")
        if self.walk_form not in (None, "auto"):
            with walk_form(self.walk_form):
                ret = _RasterizeGaussians.apply(means3D, means2D, colors_precomp, shs, opacities, scales, rotations,
                                                self.raster_settings, flavour, use_filter, cov3D_precomp)
        else:
            ret = _RasterizeGaussians.apply(means3D, means2D, colors_precomp, shs, opacities, scales, rotations,
                                            self.raster_settings, flavour, use_filter, cov3D_precomp)
        if flavour.extras:
            # how many Gaussians the ids of point_id_pixel index: lets log_amd.counter's stand-in for the
            # torch.unique call at LoG/render/renderer.py:156 recognise the map and take the histogram kernel
            ret[2]._lograst_num_gaussians = int(means3D.shape[0])
        return ret


class UpstreamGaussianRasterizer(GaussianRasterizer):
    FLAVOUR = UPSTREAM


def keep_keys(enabled):
    """Test/debug switch: forwards keep their (depth, id) key buffer in `saved` (16 bytes per tile instance, otherwise
    released when the forward returns) so that finish_lists can order the lists' tails.  Returns the previous setting."""
    global _keep_keys
    prev, _keep_keys = _keep_keys, bool(enabled)
    return prev


def ordered_lengths_of(saved, width, height):
    """Test/debug accessor: per tile, how many leading positions of its list are in final order (lists of more than
    4096 keys are ordered over their first window only unless a pixel needed more: include/lograst.h,
    lograst_ordered_lengths)."""
    tiles = ((int(width) + 15) // 16) * ((int(height) + 15) // 16)
    st = saved["state"]
    out = torch.empty(tiles, dtype=torch.int32, device=st.device)
    with torch.cuda.device(st.device):
        _lib.check(_lib.lib().lograst_ordered_lengths(_ptr(st), int(width), int(height), _ptr(out), _stream_ptr(st.device)))
    return out


def finish_lists(saved, width, height):
    """Test/debug: orders every tile list of a forward made under keep_keys(True) to its end, in place
    (lograst_finish_lists): saved["plist"] is then what LOGRAST_LAZY_SORT=0 would have produced."""
    st = saved["state"]
    if saved.get("keys") is None:
        raise RuntimeError("finish_lists: this forward did not keep its key buffer -- run it under keep_keys(True) "
                           "(the forward treats `keys` as dead scratch otherwise)")
    with torch.cuda.device(st.device):
        _lib.check(_lib.lib().lograst_finish_lists(_ptr(st), int(width), int(height), _ptr(saved["keys"]), _ptr(saved["plist"]),
                                                  int(saved["capacity"]), _stream_ptr(st.device)))


def tile_offsets_of(saved, width, height):
    """Test/debug accessor: the per-tile exclusive offsets (tiles+1 entries) inside a tile_state tensor
    (layout: log_amd/csrc/common.hpp)."""
    gx, gy = (int(width) + 15) // 16, (int(height) + 15) // 16
    tiles = gx * gy
    st = saved["state"]
    L = _lib.lib()
    first = (L.lograst_tile_offsets(ctypes.c_void_p(st.data_ptr()), int(width), int(height)) - st.data_ptr()) // 4
    return st[first:first + tiles + 1]

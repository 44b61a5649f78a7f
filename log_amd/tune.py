"""Run-time calibration of liblograst's performance knobs (include/lograst.h: lograst_set_knob).

The launch-shape parameters of the kernels -- from which size the helper passes pay for their launches, which rects go
to the wave-per-rect counting kernel, how many projection batches a workgroup owns, streaming policies of the fill, the
form of the reverse walk -- were fixed from measurements on two synthetic inputs.  ``tune()`` measures them on THIS
device with the caller's own views (or two synthetic ones: a fog of tiny splats and a heavy-tailed mix), one knob at a
time (coordinate descent over a short candidate list), keeps what is faster, stores the result per device and applies
it; ``load()`` applies a stored result.  No knob changes a result (tests/test_gpu_knobs.py sweeps them bit for bit), so
tuning is always safe to skip: the defaults are the values measured on the MI355X this library was written on.
"""
import ctypes
import json
import os
import time

from . import _lib

CANDIDATES = {
    "LOGRAST_DEFER_TILES": (8, 16, 32, 64),
    "LOGRAST_HUGE_CHUNK": (256, 512, 1024),
    "LOGRAST_BATCH_PLANES": (1, 2, 4),
    "LOGRAST_FILL_NT": (0, 1),
    "LOGRAST_FILL_XCD_ORDER": (0, 1),
    "LOGRAST_FILL_PER_THREAD": (1, 2, 4),
    "LOGRAST_BWD_ROWS": (0, 1, 2),
    "LOGRAST_FWD_ROWS": (0, 1, 2),
    "LOGRAST_BWD_BLOCK_TEST": (0, 1),
    "LOGRAST_FWD_BLOCK_TEST": (0, 1),
    "LOGRAST_MID_COOP": (0, 8, 16, 32),       # rects of 5..16 tiles: wave-cooperative counting up to this many per wave
    "LOGRAST_MID_RANK": (0, 1),
    "LOGRAST_HIT_MASKS": (0, 1),              # the forward's support ballots handed to the reverse walk
    "LOGRAST_LAZY_SORT": (0, 1),              # long lists ordered over their first window only (loses where every walk needs the tails: fog)
}
HELPER_KNOBS = ("LOGRAST_HELPER_MIN_N",)      # thresholds on the input size: tuned by helper_threshold()


def knobs():
    """[{name, default, lo, hi, what, value}] of every knob the library exposes."""
    L = _lib.lib()
    out = []
    for i in range(L.lograst_knob_count()):
        name, what = ctypes.c_char_p(), ctypes.c_char_p()
        d, lo, hi = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _lib.check(L.lograst_knob_info(i, ctypes.byref(name), ctypes.byref(d), ctypes.byref(lo), ctypes.byref(hi),
                                       ctypes.byref(what)))
        out.append(dict(name=name.value.decode(), default=d.value, lo=lo.value, hi=hi.value, what=what.value.decode(),
                        value=get_knob(name.value.decode())))
    return out


def set_knob(name, value):
    _lib.check(_lib.lib().lograst_set_knob(name.encode(), int(value)))


def get_knob(name):
    v = ctypes.c_int32()
    _lib.check(_lib.lib().lograst_get_knob(name.encode(), ctypes.byref(v)))
    return int(v.value)


def reset_knobs():
    _lib.check(_lib.lib().lograst_reset_knobs())


def cache_file(device=None):
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    p = torch.cuda.get_device_properties(dev)
    tag = "%s_%dcu" % (p.name.replace(" ", "_").replace("/", "_"), p.multi_processor_count)
    root = os.environ.get("LOGRAST_TUNE_DIR") or os.path.join(os.path.expanduser("~"), ".cache", "log_amd")
    return os.path.join(root, "tune_%s.json" % tag)


def load(path=None, device=None):
    """Apply a stored tuning (returns it, or None when there is none for this device)."""
    path = path or cache_file(device)
    try:
        with open(path) as f:
            values = json.load(f)["knobs"]
    except (OSError, ValueError, KeyError):
        return None
    known = {k["name"] for k in knobs()}
    for name, value in values.items():
        if name in known:
            set_knob(name, value)
    return values


def synthetic_views(device, n=1_000_000, width=1920, height=1080, heavy_tail=False):
    """One callable running forward + backward of a synthetic view through the drop-in package: `n` random Gaussians in
    the unit cube seen from 3 units away (the bench scenes); heavy_tail: 2 % of them 20x larger (rects of tens of tiles,
    what a level-of-detail selection hands over)."""
    import math
    import numpy as np
    import torch
    from . import scenes
    from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device(device)
    gen = torch.Generator(device=dev).manual_seed(7 if heavy_tail else 3)
    smax = 0.5 * float(n) ** (-1.0 / 3.0)
    scales = torch.rand(n, 3, device=dev, generator=gen) * smax
    if heavy_tail:
        big = torch.rand(n, device=dev, generator=gen) < 0.02
        scales = torch.where(big[:, None], scales * 20.0, scales)
    leaves = dict(means3D=torch.rand(n, 3, device=dev, generator=gen) - 0.5, scales=scales,
                  rotations=torch.nn.functional.normalize(torch.rand(n, 4, device=dev, generator=gen) + 1e-3),
                  opacities=torch.rand(n, 1, device=dev, generator=gen), colors=torch.rand(n, 3, device=dev, generator=gen))
    leaves = {k: v.requires_grad_(True) for k, v in leaves.items()}
    cam = scenes.orbit_cameras(8, W=width, H=height, focal=2139.0 * width / 1920.0)[1]
    T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
    rs = GaussianRasterizationSettings(
        image_height=height, image_width=width, tanfovx=math.tan(cam["FoVx"] * 0.5), tanfovy=math.tan(cam["FoVy"] * 0.5),
        bg=T([1.0, 1.0, 1.0]), scale_modifier=1.0, viewmatrix=T(cam["world_view_transform"]),
        projmatrix=T(cam["full_proj_transform"]), sh_degree=0, campos=T(cam["camera_center"]), prefiltered=False, debug=False)
    rast = GaussianRasterizer(raster_settings=rs)
    w = torch.rand(3, height, width, device=dev, generator=gen)

    def view():
        for v in leaves.values():
            v.grad = None
        m2 = torch.empty(n, 3, device=dev).requires_grad_(True)
        out = rast(means3D=leaves["means3D"], means2D=m2, shs=None, colors_precomp=leaves["colors"],
                   opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
        out[0].backward(gradient=w)
    return view


def _time(views, repeats):
    import torch
    best = float("inf")
    for v in views:          # warm-up: allocator, capacity history
        v()
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for v in views:
            v()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def helper_threshold(device, repeats=3, sizes=(2_000_000, 6_000_000)):
    """LOGRAST_HELPER_MIN_N: the helper passes (absolute slot table, touched-only clearing, separate zero-fills) on or off
    at two input sizes; the threshold goes below the smallest size at which they win."""
    result = 16_000_000
    for n in sorted(sizes, reverse=True):
        view = synthetic_views(device, n=n)
        set_knob("LOGRAST_HELPER_MIN_N", 0)
        on = _time([view], repeats)
        set_knob("LOGRAST_HELPER_MIN_N", 2_000_000_000)
        off = _time([view], repeats)
        if on < off:
            result = n // 2
        else:
            break
    set_knob("LOGRAST_HELPER_MIN_N", result)
    return result


def tune(views=None, device=None, repeats=3, candidates=None, save=True, path=None, helper=True, margin=0.01, log=None):
    """Calibrate the knobs on this device.  views: callables that each run one forward + backward view of the caller's
    workload (default: the two synthetic views above).  A candidate replaces the current value only when it is faster by
    more than `margin` (timing noise must not move a default).  -> {knob: value}; stored for load() when `save`."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if views is None:
        views = [synthetic_views(dev, 1_000_000), synthetic_views(dev, 2_000_000, heavy_tail=True)]
    cands = dict(CANDIDATES if candidates is None else candidates)
    chosen, report = {}, {}
    with torch.cuda.device(dev):
        base = _time(views, repeats)
        for name, values in cands.items():
            start = get_knob(name)
            best_v, best_t, seen = start, base, {start: base}
            for val in values:
                if val == start:
                    continue
                set_knob(name, val)
                seen[val] = _time(views, repeats)
                if seen[val] < best_t * (1.0 - margin):
                    best_v, best_t = val, seen[val]
            set_knob(name, best_v)
            chosen[name], base = best_v, best_t
            report[name] = {str(k): round(1e3 * t, 4) for k, t in seen.items()}
            if log:
                log("%s -> %d  (ms per pass: %s)" % (name, best_v, report[name]))
        if helper:
            chosen["LOGRAST_HELPER_MIN_N"] = helper_threshold(dev, repeats)
    if save:
        path = path or cache_file(dev)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump({"knobs": chosen, "timings_ms": report, "device": torch.cuda.get_device_properties(dev).name}, f, indent=1)
    return chosen

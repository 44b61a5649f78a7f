"""ctypes/numpy front-end of the CPU oracle (oracle/lograst_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  Nothing under ``log_amd/`` imports it.  Parity status and the reference file:line each
stage follows are documented in the header of ``lograst_oracle.c``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblograst_oracle.so")

FILTER_NONE, FILTER_DILATE, FILTER_CLAMP = 0, 1, 2
REC = 12
TILE = 16


class OraView(ctypes.Structure):
    _fields_ = [
        ("width", ctypes.c_int32), ("height", ctypes.c_int32),
        ("tanfovx", ctypes.c_float), ("tanfovy", ctypes.c_float),
        ("scale_modifier", ctypes.c_float),
        ("filter_mode", ctypes.c_int32), ("ndc_cull", ctypes.c_int32),
        ("view", ctypes.c_float * 16), ("proj", ctypes.c_float * 16), ("bg", ctypes.c_float * 3),
    ]


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "lograst_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liblograst_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.ora_project.restype = ctypes.c_int64
    return _lib


def _p(a, t=None):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def make_view(width, height, tanfovx, tanfovy, viewmatrix, projmatrix, bg, scale_modifier=1.0,
              filter_mode=FILTER_CLAMP, ndc_cull=1):
    v = OraView()
    v.width, v.height = int(width), int(height)
    v.tanfovx, v.tanfovy = float(tanfovx), float(tanfovy)
    v.scale_modifier = float(scale_modifier)
    v.filter_mode, v.ndc_cull = int(filter_mode), int(ndc_cull)
    v.view[:] = _f32(viewmatrix).reshape(-1).tolist()
    v.proj[:] = _f32(projmatrix).reshape(-1).tolist()
    v.bg[:] = _f32(bg).reshape(-1).tolist()
    return v


def grid(view):
    gx = (view.width + TILE - 1) // TILE
    gy = (view.height + TILE - 1) // TILE
    return gx, gy


def compute_radius(means, scales, rots, proj, viewm, fx, fy, tanfovx, tanfovy):
    """A0 -- LoG/cuda/compute_radius_kernel.cu:107-183."""
    means, scales, rots = _f32(means), _f32(scales), _f32(rots)
    proj, viewm = _f32(proj).reshape(-1), _f32(viewm).reshape(-1)
    P = means.shape[0]
    out = np.zeros(P, np.float32)
    lib().ora_compute_radius(ctypes.c_int32(P), _p(means), _p(scales), _p(rots), _p(proj), _p(viewm),
                             ctypes.c_float(fx), ctypes.c_float(fy), ctypes.c_float(tanfovx),
                             ctypes.c_float(tanfovy), _p(out))
    return out


def lod_radius(idx, xyz, scaling, rotation, proj, viewm, fx, fy, tanfovx, tanfovy):
    """Gaussian.compute_radius for TensorTree.traverse (LoG/model/level_of_gaussian.py:65-88) on RAW parameters:
    gather `idx`, exp / normalize, A0."""
    idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int64))
    xyz, scaling, rotation = _f32(xyz), _f32(scaling), _f32(rotation)
    proj, viewm = _f32(proj).reshape(-1), _f32(viewm).reshape(-1)
    out = np.zeros(idx.shape[0], np.float32)
    lib().ora_lod_radius(ctypes.c_int32(idx.shape[0]), _p(idx), _p(xyz), _p(scaling), _p(rotation), _p(proj),
                         _p(viewm), ctypes.c_float(fx), ctypes.c_float(fy), ctypes.c_float(tanfovx),
                         ctypes.c_float(tanfovy), _p(out))
    return out


def lod_traverse(node_index, tree, xyz, scaling, rotation, root_index, proj, viewm, fx, fy, tanfovx, tanfovy,
                 min_resolution_pixel, max_level, max_depth=1000):
    """N3 -- TensorTree.traverse + _query_tree_torch, LoG/model/tensor_tree.py:131-185, restated with numpy.
    Returns the selected point indices (int64) in the reference's order."""
    node_index = np.asarray(node_index, dtype=np.int64)
    tree = np.asarray(tree, dtype=np.int64).reshape(-1, tree.shape[-1] if np.ndim(tree) == 2 else 1)
    root_index = np.asarray(root_index, dtype=np.int64)
    min_px = np.float32(min_resolution_pixel)

    def radius(idx):
        return lod_radius(idx, xyz, scaling, rotation, proj, viewm, fx, fy, tanfovx, tanfovy)

    # tensor_tree.py:167-174
    keep = (radius(root_index) < min_px) | (node_index[root_index] == -1)
    out = [root_index[keep]]
    index = root_index[~keep]
    # tensor_tree.py:131-165
    level = 1
    while True:
        if level > max_level or level > max_depth:
            out.append(index)
            break
        child = tree[node_index[index]].reshape(-1)
        child = child[child != -1]
        keep = (radius(child) < min_px) | (node_index[child] == -1)
        out.append(child[keep])
        if not (~keep).any():
            break
        index = child[~keep]
        level += 1
    return np.concatenate(out).astype(np.int64)


def forward(view, means, scales, rots, opac, colors, extras=True, cov3d=None, tile_rows=None):
    """Full forward.  Returns a dict with every intermediate the HIP path is compared against.
    cov3d ([N, 6], the rasterizer's cov3D_precomp) replaces scales + rots when given.
    tile_rows=(begin, end): the whole-view render RESTRICTED to the band of tile rows [begin, end) -- what one rank of an
    image split across GPUs owns (include/lograst.h: lograst_view.tile_row_begin / _end): after the
    whole-view projection every rect is clipped to the band's rows, a Gaussian whose clipped rect is empty is dropped
    (radii = 0, no record), and binning / compositing then see the band's tiles only -- every tile outside the band has an
    empty list (image = background there).  The per-Gaussian arithmetic is the whole view's, untouched."""
    means = _f32(means)
    N = means.shape[0]
    cov3d = None if cov3d is None else _f32(cov3d).reshape(N, 6)
    scales = _f32(scales) if cov3d is None else np.zeros((N, 3), np.float32)
    rots = _f32(rots) if cov3d is None else np.zeros((N, 4), np.float32)
    opac, colors = _f32(opac).reshape(-1), _f32(colors)
    W, H = view.width, view.height
    gx, gy = grid(view)
    T = gx * gy
    radii = np.zeros(N, np.int32)
    rec = np.zeros((max(N, 1), REC), np.float32)
    touched = np.zeros(max(N, 1), np.uint32)
    L = lib()
    I = L.ora_project(ctypes.byref(view), ctypes.c_int32(N), _p(means), _p(scales), _p(rots), _p(opac),
                      _p(colors), _p(radii), _p(rec), _p(touched), _p(cov3d) if cov3d is not None else None)
    if tile_rows is not None and tuple(tile_rows) != (0, 0):
        b, e = int(tile_rows[0]), int(tile_rows[1])
        r0, r1 = rec[:N, 10].view(np.uint32), rec[:N, 11].view(np.uint32)
        x0, x1 = r0 & 0xffff, r1 & 0xffff
        y0 = np.clip(r0 >> 16, b, e).astype(np.uint32)
        y1 = np.clip(r1 >> 16, b, e).astype(np.uint32)
        keep = (radii > 0) & (y1 > y0)
        radii[~keep] = 0
        touched[:N] = np.where(keep, (x1 - x0) * (y1 - y0), 0).astype(np.uint32)
        r0[:] = np.where(keep, x0 | (y0 << 16), 0)
        r1[:] = np.where(keep, x1 | (y1 << 16), 0)
        I = int(touched[:N].astype(np.int64).sum())
        del x0, x1, y0, y1, keep
    offsets = np.zeros(T + 1, np.uint32)
    plist = np.zeros(max(int(I), 1), np.uint32)
    rc = L.ora_bin(ctypes.byref(view), ctypes.c_int32(N), _p(radii), _p(rec), _p(offsets), _p(plist))
    assert rc == 0
    image = np.zeros((3, H, W), np.float32)
    final_T = np.zeros((H, W), np.float32)
    n_contrib = np.zeros((H, W), np.int32)
    pid = np.full((H, W), -1, np.int32) if extras else None
    pwp = np.zeros((H, W), np.float32) if extras else None
    pw = np.zeros(max(N, 1), np.float32) if extras else None
    L.ora_blend_fwd(ctypes.byref(view), ctypes.c_int32(N), _p(rec), _p(offsets), _p(plist), _p(image),
                    _p(final_T), _p(n_contrib),
                    _p(pid) if extras else None, _p(pwp) if extras else None, _p(pw) if extras else None)
    return dict(N=N, I=int(I), radii=radii, rec=rec[:N], tiles_touched=touched[:N], tile_offsets=offsets,
                point_list=plist[:int(I)], image=image, final_T=final_T, n_contrib=n_contrib,
                point_id_pixel=pid, point_weight_pixel=pwp, point_weight=(pw[:N] if extras else None),
                inputs=(means, scales, rots, opac, colors), cov3d=cov3d)


def instance_support(view, fwd):
    """uint8[I]: 1 where the tile instance can pass the alpha floor at some pixel of its tile (see the C header)."""
    flags = np.zeros(max(fwd["I"], 1), np.uint8)
    lib().ora_instance_support(ctypes.byref(view), _p(fwd["rec"]), _p(fwd["tile_offsets"]), _p(fwd["point_list"]),
                               _p(flags))
    return flags[:fwd["I"]]


def backward(view, fwd, dL_dimage):
    """Full backward for a forward() result.  Returns grads dict."""
    means, scales, rots, opac, colors = fwd["inputs"]
    N = fwd["N"]
    dL = _f32(dL_dimage)
    n = max(N, 1)
    g_mean2d = np.zeros((n, 3), np.float32)
    g_conic = np.zeros((n, 4), np.float32)
    g_opac = np.zeros(n, np.float32)
    g_col = np.zeros((n, 3), np.float32)
    g_means = np.zeros((n, 3), np.float32)
    g_scales = np.zeros((n, 3), np.float32)
    g_rots = np.zeros((n, 4), np.float32)
    L = lib()
    rec = np.ascontiguousarray(fwd["rec"]) if N else np.zeros((1, REC), np.float32)
    plist = fwd["point_list"] if fwd["I"] else np.zeros(1, np.uint32)
    L.ora_blend_bwd(ctypes.byref(view), ctypes.c_int32(N), _p(rec), _p(fwd["tile_offsets"]), _p(plist),
                    _p(fwd["final_T"]), _p(fwd["n_contrib"]), _p(dL), _p(g_mean2d), _p(g_conic), _p(g_opac),
                    _p(g_col))
    cov3d = fwd.get("cov3d")
    g_cov = np.zeros((n, 6), np.float32) if cov3d is not None else None
    L.ora_project_bwd(ctypes.byref(view), ctypes.c_int32(N), _p(means), _p(scales), _p(rots),
                      _p(fwd["radii"]), _p(g_mean2d), _p(g_conic), _p(g_means), _p(g_scales), _p(g_rots),
                      _p(cov3d) if cov3d is not None else None, _p(g_cov) if cov3d is not None else None)
    out = dict(means3D=g_means[:N], means2D=g_mean2d[:N], scales=g_scales[:N], rotations=g_rots[:N],
               opacities=g_opac[:N].reshape(-1, 1), colors=g_col[:N], conic=g_conic[:N])
    if cov3d is not None:
        out["cov3D"] = g_cov[:N]
    return out


def backward_f64(view, fwd, dL_dimage, cond=True, pert=1e-7, trials=3):
    """The float64 twin of backward() at full size (C: ora64_blend_bwd + ora64_project_bwd): every value in double, every
    decision the fp32 forward's.  Returns the same keys as backward() as float64 arrays, plus (cond=True) "cond":
    float64[N, 3] = how much the chain rule amplifies a relative perturbation of its inputs, for the means3D / scales /
    rotations row of every Gaussian (see the C header).  Not available with cov3D_precomp."""
    assert fwd.get("cov3d") is None, "the float64 twin covers the scales / rotations path"
    means, scales, rots, opac, colors = fwd["inputs"]
    N = fwd["N"]
    n = max(N, 1)
    dL = _f32(dL_dimage)
    z = lambda *shape: np.zeros(shape, np.float64)
    g_mean2d, g_conic, g_opac, g_col = z(n, 3), z(n, 4), z(n), z(n, 3)
    g_means, g_scales, g_rots = z(n, 3), z(n, 3), z(n, 4)
    cnd = z(n, 3) if cond else None
    g_abs = z(n, 5) if cond else None
    L = lib()
    rec = np.ascontiguousarray(fwd["rec"]) if N else np.zeros((1, REC), np.float32)
    plist = fwd["point_list"] if fwd["I"] else np.zeros(1, np.uint32)
    L.ora64_blend_bwd(ctypes.byref(view), ctypes.c_int32(N), _p(rec), _p(fwd["tile_offsets"]), _p(plist),
                      _p(fwd["n_contrib"]), _p(dL), _p(g_mean2d), _p(g_conic), _p(g_opac), _p(g_col),
                      _p(g_abs) if cond else None)
    L.ora64_project_bwd(ctypes.byref(view), ctypes.c_int32(N), _p(means), _p(scales), _p(rots), _p(fwd["radii"]),
                        _p(g_mean2d), _p(g_conic), _p(g_abs) if cond else None, _p(g_means), _p(g_scales), _p(g_rots),
                        _p(cnd) if cond else None, ctypes.c_double(pert), ctypes.c_int32(trials))
    out = dict(means3D=g_means[:N], means2D=g_mean2d[:N], scales=g_scales[:N], rotations=g_rots[:N],
               opacities=g_opac[:N].reshape(-1, 1), colors=g_col[:N], conic=g_conic[:N])
    if cond:
        out["cond"] = cnd[:N]
        # the fp32 chain rule (ora_project_bwd: the op sequence the HIP kernel shares) on the float64 sums rounded to fp32:
        # its distance from the float64 chain rule is the EVALUATION error of that row in fp32 -- for a needle-shaped
        # Gaussian (two scales a thousand times below the third) the rounding of intermediates that perturbing the inputs
        # cannot reach (the six entries of Sigma rounded independently, det = ac - b^2 of a nearly rank-1 covariance)
        m32, s32, r32 = (np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 4), np.float32))
        g2_32 = np.ascontiguousarray(g_mean2d, np.float32)
        gc_32 = np.ascontiguousarray(g_conic, np.float32)
        L.ora_project_bwd(ctypes.byref(view), ctypes.c_int32(N), _p(means), _p(scales), _p(rots), _p(fwd["radii"]),
                          _p(g2_32), _p(gc_32), _p(m32), _p(s32), _p(r32), None, None)
        out["chain32"] = dict(means3D=m32[:N], scales=s32[:N], rotations=r32[:N])
    return out


def project_backward(view, fwd, g_mean2d, g_conic):
    """A6b alone: chain rule from (dL/d ndc-mean [N,3], dL/d conic [N,4]) to means3D/scales/rotations."""
    means, scales, rots, opac, colors = fwd["inputs"]
    N = fwd["N"]
    n = max(N, 1)
    gm2 = np.zeros((n, 3), np.float32); gm2[:N] = g_mean2d
    gc = np.zeros((n, 4), np.float32); gc[:N] = g_conic
    g_means = np.zeros((n, 3), np.float32)
    g_scales = np.zeros((n, 3), np.float32)
    g_rots = np.zeros((n, 4), np.float32)
    cov3d = fwd.get("cov3d")
    g_cov = np.zeros((n, 6), np.float32) if cov3d is not None else None
    lib().ora_project_bwd(ctypes.byref(view), ctypes.c_int32(N), _p(means), _p(scales), _p(rots),
                          _p(fwd["radii"]), _p(gm2), _p(gc), _p(g_means), _p(g_scales), _p(g_rots),
                          _p(cov3d) if cov3d is not None else None, _p(g_cov) if cov3d is not None else None)
    out = dict(means3D=g_means[:N], scales=g_scales[:N], rotations=g_rots[:N])
    if cov3d is not None:
        out["cov3D"] = g_cov[:N]
    return out


def sh_forward(means, campos, shs, degree):
    """N2: colours[N,3], clamped[N,3] for shs[N,M,3]."""
    means, campos, shs = _f32(means), _f32(campos).reshape(-1), _f32(shs)
    N, M = shs.shape[0], shs.shape[1]
    colors = np.zeros((max(N, 1), 3), np.float32)
    clamped = np.zeros((max(N, 1), 3), np.uint8)
    lib().ora_sh_fwd(ctypes.c_int32(N), ctypes.c_int32(degree), ctypes.c_int32(M), _p(means), _p(campos), _p(shs),
                     _p(colors), _p(clamped))
    return colors[:N], clamped[:N]


def sh_backward(means, campos, shs, degree, clamped, g_colors):
    """N2 backward: (dL/dshs[N,M,3], dL/dmeans3D contribution [N,3])."""
    means, campos, shs = _f32(means), _f32(campos).reshape(-1), _f32(shs)
    N, M = shs.shape[0], shs.shape[1]
    g_colors = _f32(g_colors)
    cl = np.ascontiguousarray(clamped, np.uint8)
    g_shs = np.zeros((max(N, 1), M, 3), np.float32)
    g_means = np.zeros((max(N, 1), 3), np.float32)
    lib().ora_sh_bwd(ctypes.c_int32(N), ctypes.c_int32(degree), ctypes.c_int32(M), _p(means), _p(campos), _p(shs),
                     _p(cl), _p(g_colors), _p(g_shs), _p(g_means))
    return g_shs[:N], g_means[:N]


def id_histogram(point_id_pixel):
    """N4a -- LoG/render/renderer.py:156-159: sorted distinct ids (without -1) and how many pixels each one wins."""
    ids, counts = np.unique(np.asarray(point_id_pixel).reshape(-1), return_counts=True)
    if ids.size and ids[0] == -1:
        ids, counts = ids[1:], counts[1:]
    return ids.astype(np.int32), counts.astype(np.int64)


COUNTER_FIELDS = (("weights_max", np.float32), ("weights_sum", np.float32), ("grad_sum", np.float32),
                  ("radii_max", np.int16), ("visible_count", np.int16), ("radii_max_max", np.int32),
                  ("area_sum", np.int32), ("create_steps", np.int32))


def counter_update(state, visible_index, grad, radii, point_weight, point_id, point_count):
    """N4b -- Counter.update_by_output for one view (LoG/model/counter.py:36-68).  `state`: dict of the eight
    Counter buffers (COUNTER_FIELDS), updated IN PLACE (arrays must be contiguous, of the registered dtypes).
    Returns flag_vis (bool[nv])."""
    vi = np.ascontiguousarray(np.asarray(visible_index, np.int64))
    g = _f32(grad)
    r = np.ascontiguousarray(np.asarray(radii, np.int32))
    w = _f32(point_weight)
    pid = np.ascontiguousarray(np.asarray(point_id, np.int32))
    pc = np.ascontiguousarray(np.asarray(point_count, np.int64))
    for name, dt in COUNTER_FIELDS:
        assert state[name].dtype == dt and state[name].flags["C_CONTIGUOUS"], name
    flag = np.zeros(vi.shape[0], np.uint8)
    lib().ora_counter_update(ctypes.c_int32(vi.shape[0]), _p(vi), _p(g), _p(r), _p(w), ctypes.c_int32(pid.shape[0]),
                             _p(pid), _p(pc), *[_p(state[name]) for name, _ in COUNTER_FIELDS], _p(flag))
    return flag.astype(bool)


def sparse_adam(model, param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, index, flag_vis, lr, step, beta1=0.9,
                beta2=0.999, eps=1e-15):
    """N4c -- one key of SparseOptimizer.step (LoG/model/sparse_optimizer.py:41-78,163-196).  model / exp_avg /
    exp_avg_sq / max_exp_avg_sq ([P, ...] fp32, contiguous) are updated IN PLACE; the Python-side scalars follow
    sparse_optimizer.py:62-71."""
    import math
    idx = np.ascontiguousarray(np.asarray(index, np.int64))
    fv = np.ascontiguousarray(np.asarray(flag_vis, np.uint8))
    param, grad = _f32(param), _f32(grad)
    width = int(np.prod(param.shape[1:])) if param.ndim > 1 else 1
    bc1 = 1 - beta1 ** int(step)
    bc2 = 1 - beta2 ** int(step)
    for a in (model, exp_avg, exp_avg_sq):
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    lib().ora_sparse_adam(ctypes.c_int32(idx.shape[0]), _p(idx), _p(fv), ctypes.c_int32(width), _p(model), _p(param),
                          _p(grad), _p(exp_avg), _p(exp_avg_sq),
                          _p(max_exp_avg_sq) if max_exp_avg_sq is not None else ctypes.c_void_p(0),
                          ctypes.c_double(lr / bc1), ctypes.c_double(beta1), ctypes.c_double(beta2),
                          ctypes.c_double(math.sqrt(bc2)), ctypes.c_double(eps))


def gather_activate(index, xyz, scaling, opacity, rotation, colors, shs, degree, campos):
    """Rows N2/N3 -- LoG.get_all + Activation.activate_root_return (LoG/model/level_of_gaussian.py:262-296,
    LoG/model/activation.py:27-44).  shs: [P,K,3] or None.  -> (raw dict, activated dict)."""
    idx = np.ascontiguousarray(np.asarray(index, np.int64))
    n = idx.shape[0]
    xyz, scaling, opacity, rotation, colors = (_f32(a) for a in (xyz, scaling, opacity, rotation, colors))
    K = 0 if shs is None else int(shs.shape[1])
    shs_c = _f32(shs) if K else np.zeros((1, 1, 3), np.float32)
    cp = _f32(campos if campos is not None else np.zeros(3)).reshape(-1)
    raw = {"xyz": np.zeros((n, 3), np.float32), "scaling": np.zeros((n, 3), np.float32),
           "opacity": np.zeros((n, 1), np.float32), "rotation": np.zeros((n, 4), np.float32),
           "colors": np.zeros((n, 3), np.float32), "shs": np.zeros((n, K, 3), np.float32)}
    act = {"scaling": np.zeros((n, 3), np.float32), "opacity": np.zeros((n, 1), np.float32),
           "rotation": np.zeros((n, 4), np.float32), "colors": np.zeros((n, 3), np.float32)}
    lib().ora_gather_activate(ctypes.c_int32(n), _p(idx), _p(xyz), _p(scaling), _p(opacity), _p(rotation), _p(colors),
                              _p(shs_c), ctypes.c_int32(K), ctypes.c_int32(int(degree)), _p(cp), _p(raw["xyz"]),
                              _p(raw["scaling"]), _p(raw["opacity"]), _p(raw["rotation"]), _p(raw["colors"]),
                              _p(raw["shs"]), _p(act["scaling"]), _p(act["opacity"]), _p(act["rotation"]),
                              _p(act["colors"]))
    act["xyz"] = raw["xyz"]
    if not K:
        del raw["shs"]
    return raw, act


def activate_backward(raw, degree, campos, g_scaling, g_opacity, g_rotation, g_colors):
    """Gradients of the raw rows from the gradients of the activated tensors (same row count)."""
    n = raw["xyz"].shape[0]
    K = int(raw["shs"].shape[1]) if "shs" in raw else 0
    cp = _f32(campos if campos is not None else np.zeros(3)).reshape(-1)
    g = {"scaling": np.zeros((n, 3), np.float32), "opacity": np.zeros((n, 1), np.float32),
         "rotation": np.zeros((n, 4), np.float32), "colors": np.zeros((n, 3), np.float32)}
    if K:
        g["shs"] = np.zeros((n, K, 3), np.float32)
    lib().ora_activate_backward(ctypes.c_int32(n), _p(_f32(raw["xyz"])), _p(_f32(raw["scaling"])),
                                _p(_f32(raw["opacity"])), _p(_f32(raw["rotation"])), ctypes.c_int32(K),
                                ctypes.c_int32(int(degree)), _p(cp), _p(_f32(g_scaling)), _p(_f32(g_opacity)),
                                _p(_f32(g_rotation)), _p(_f32(g_colors)), _p(g["scaling"]), _p(g["opacity"]),
                                _p(g["rotation"]), _p(g["colors"]), _p(g["shs"]) if K else ctypes.c_void_p(0))
    return g

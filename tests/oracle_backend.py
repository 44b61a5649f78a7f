"""TEST DOUBLE: a CPU backend for log_amd.rasterizer built on the oracle, so that the reference's unmodified
Python (LoG/render/renderer.py, LoG/model/*) can be driven end-to-end on a machine without a GPU
(BASELINE config C1, "plumbing").  Installed only by tests, through install() below (it swaps the module attribute
log_amd.rasterizer._backend; the product module has no hook for this); the product path never imports this file."""
import math

import numpy as np
import torch

from oracle import oracle
from log_amd import _lib


def install(backend):
    """Swap log_amd.rasterizer's backend object (None = a fresh HipBackend); returns the previous one."""
    from log_amd import rasterizer as R
    old = R._backend
    R._backend = backend if backend is not None else R.HipBackend()
    return old


class OracleBackend:
    def _view(self, rs, flavour, use_filter):
        fm = flavour.filter_mode if use_filter else _lib.FILTER_NONE
        return oracle.make_view(int(rs.image_width), int(rs.image_height), float(rs.tanfovx), float(rs.tanfovy),
                                rs.viewmatrix.detach().cpu().numpy(), rs.projmatrix.detach().cpu().numpy(),
                                rs.bg.detach().cpu().numpy(), scale_modifier=float(rs.scale_modifier),
                                filter_mode=fm, ndc_cull=flavour.ndc_cull)

    def forward(self, rs, flavour, use_filter, means3D, scales, rotations, opacities, colors, scratch_floats=0,
                cov3D=None):
        v = self._view(rs, flavour, use_filter)
        f = oracle.forward(v, means3D.numpy(), None if scales is None else scales.numpy(),
                           None if rotations is None else rotations.numpy(), opacities.numpy(), colors.numpy(),
                           extras=bool(flavour.extras), cov3d=None if cov3D is None else cov3D.numpy())
        t = torch.from_numpy
        image, radii = t(f["image"]), t(f["radii"])
        if flavour.extras:
            return image, radii, t(f["point_id_pixel"]), t(f["point_weight_pixel"]), t(f["point_weight"].copy()), (v, f)
        return image, radii, None, None, None, (v, f)

    def backward(self, rs, flavour, use_filter, means3D, scales, rotations, saved, grad_image, sink=None, cov3D=None):
        assert sink is None, 'the CPU test double does not implement the accumulate path'
        v, f = saved
        g = oracle.backward(v, f, grad_image.detach().cpu().numpy())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        if cov3D is not None:
            return (t(g["means3D"]), t(g["means2D"]), t(g["colors"]), t(g["opacities"].reshape(-1)), t(g["cov3D"]), None)
        return (t(g["means3D"]), t(g["means2D"]), t(g["colors"]), t(g["opacities"].reshape(-1)), t(g["scales"]),
                t(g["rotations"]))

    def compute_radius(self, means3D, scales, rotations, projmatrix, viewmatrix, fx, fy, tanfovx, tanfovy):
        n = lambda a: a.detach().cpu().numpy()
        r = oracle.compute_radius(n(means3D), n(scales), n(rotations), n(projmatrix), n(viewmatrix), float(fx),
                                  float(fy), float(tanfovx), float(tanfovy))
        return torch.from_numpy(r)

    def sh_forward(self, means3D, campos, shs, degree):
        c, cl = oracle.sh_forward(means3D.numpy(), campos.detach().cpu().numpy(), shs.numpy(), int(degree))
        return torch.from_numpy(c.copy()), torch.from_numpy(cl.copy())

    def sh_backward(self, means3D, campos, shs, degree, clamped, g_colors, g_means3D):
        g_shs, g_m = oracle.sh_backward(means3D.numpy(), campos.detach().cpu().numpy(), shs.numpy(), int(degree),
                                        clamped.numpy(), g_colors.numpy())
        g_means3D += torch.from_numpy(g_m.copy())
        return torch.from_numpy(g_shs.copy())

    def lod_traverse(self, node_index, tree, xyz, scaling, rotation, root_index, projmatrix, viewmatrix, fx, fy,
                     tanfovx, tanfovy, min_resolution_pixel, levels, depth_hint=None):
        n = lambda a: a.detach().cpu().numpy()
        tr = n(tree)
        idx = oracle.lod_traverse(n(node_index), tr.reshape(-1, tr.shape[-1]) if tr.ndim == 2 else tr.reshape(0, 1),
                                  n(xyz), n(scaling), n(rotation), n(root_index), n(projmatrix), n(viewmatrix),
                                  float(fx), float(fy), float(tanfovx), float(tanfovy), float(min_resolution_pixel),
                                  max_level=int(levels), max_depth=int(levels))
        return torch.from_numpy(idx)

    def id_histogram(self, point_id_pixel, n):
        ids, counts = oracle.id_histogram(point_id_pixel.detach().cpu().numpy())
        return torch.from_numpy(ids), torch.from_numpy(counts)

    def counter_update(self, buffers, visible_index, grad, radii, point_weight, point_id, point_count):
        n = lambda a: a.detach().cpu().numpy()
        state = {k: buffers[k].numpy() for k, _ in oracle.COUNTER_FIELDS}      # shares memory: updated in place
        flag = oracle.counter_update(state, n(visible_index), n(grad), n(radii), n(point_weight), n(point_id),
                                     n(point_count))
        return torch.from_numpy(flag)

    def sparse_adam(self, index, flag_vis, entries, beta1, beta2, bias_correction2_sqrt, eps):
        import ctypes
        n = lambda a: a.detach().cpu().numpy()
        idx = np.ascontiguousarray(n(index).astype(np.int64))
        fv = np.ascontiguousarray(n(flag_vis).astype(np.uint8))
        for model_p, param, grad, m1, m2, mmax, step_size in entries:
            width = int(model_p[0].numel())
            p, g = np.ascontiguousarray(n(param), np.float32), np.ascontiguousarray(n(grad), np.float32)
            ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
            oracle.lib().ora_sparse_adam(
                ctypes.c_int32(idx.shape[0]), ptr(idx), ptr(fv), ctypes.c_int32(width), ptr(model_p.numpy()), ptr(p),
                ptr(g), ptr(m1.numpy()), ptr(m2.numpy()),
                ptr(mmax.numpy()) if mmax is not None else ctypes.c_void_p(0), ctypes.c_double(step_size),
                ctypes.c_double(beta1), ctypes.c_double(beta2), ctypes.c_double(bias_correction2_sqrt),
                ctypes.c_double(eps))

    def gather_activate(self, index, bufs, degree, campos):
        n = lambda a: a.detach().cpu().numpy()
        raw, act = oracle.gather_activate(n(index), n(bufs["xyz"]), n(bufs["scaling"]), n(bufs["opacity"]),
                                          n(bufs["rotation"]), n(bufs["colors"]), n(bufs["shs"]) if "shs" in bufs else None,
                                          int(degree), n(campos) if campos is not None else None)
        t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
        raw, act = t(raw), t(act)
        act["xyz"] = raw["xyz"]
        return raw, act

    def activate_backward(self, raw, n, degree, campos, g_scaling, g_opacity, g_rotation, g_colors):
        a = lambda x: x.detach().cpu().numpy()
        sub = {k: a(v)[:n] for k, v in raw.items()}
        g = oracle.activate_backward(sub, int(degree), a(campos) if campos is not None else None, a(g_scaling)[:n],
                                     a(g_opacity)[:n], a(g_rotation)[:n], a(g_colors)[:n])
        if degree <= 0:
            g.pop("shs", None)
        return {k: torch.from_numpy(v) for k, v in g.items()}

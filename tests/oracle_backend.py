"""TEST DOUBLE: a CPU backend for log_amd.rasterizer built on the oracle, so that the reference's unmodified
Python (LoG/render/renderer.py, LoG/model/*) can be driven end-to-end on a machine without a GPU
(BASELINE config C1, "plumbing").  Installed only by tests through rasterizer._set_backend_for_tests; the
product path never imports this file."""
import math

import numpy as np
import torch

from oracle import oracle
from log_amd import _lib


class OracleBackend:
    def _view(self, rs, flavour, use_filter):
        fm = flavour.filter_mode if use_filter else _lib.FILTER_NONE
        return oracle.make_view(int(rs.image_width), int(rs.image_height), float(rs.tanfovx), float(rs.tanfovy),
                                rs.viewmatrix.detach().cpu().numpy(), rs.projmatrix.detach().cpu().numpy(),
                                rs.bg.detach().cpu().numpy(), scale_modifier=float(rs.scale_modifier),
                                filter_mode=fm, ndc_cull=flavour.ndc_cull)

    def forward(self, rs, flavour, use_filter, means3D, scales, rotations, opacities, colors, scratch_floats=0):
        v = self._view(rs, flavour, use_filter)
        f = oracle.forward(v, means3D.numpy(), scales.numpy(), rotations.numpy(), opacities.numpy(), colors.numpy(),
                           extras=bool(flavour.extras))
        t = torch.from_numpy
        image, radii = t(f["image"]), t(f["radii"])
        if flavour.extras:
            return image, radii, t(f["point_id_pixel"]), t(f["point_weight_pixel"]), t(f["point_weight"].copy()), (v, f)
        return image, radii, None, None, None, (v, f)

    def backward(self, rs, flavour, use_filter, means3D, scales, rotations, saved, grad_image, sink=None):
        assert sink is None, 'the CPU test double does not implement the accumulate path'
        v, f = saved
        g = oracle.backward(v, f, grad_image.detach().cpu().numpy())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
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

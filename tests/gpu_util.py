"""Helpers for the -m gpu parity tests: run the HIP path (through log_amd's C-ABI binding) and the CPU
oracle on the same inputs."""
import math

import numpy as np
import torch

from log_amd import _lib, rasterizer as R
from util import cam_tan


def settings(cam, bg, dev, scale_modifier=1.0):
    tfx, tfy = cam_tan(cam)
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=dev)
    return R.GaussianRasterizationSettings(
        image_height=cam["image_height"], image_width=cam["image_width"], tanfovx=tfx, tanfovy=tfy, bg=t(bg),
        scale_modifier=scale_modifier, viewmatrix=t(cam["world_view_transform"]),
        projmatrix=t(cam["full_proj_transform"]), sh_degree=0, campos=t(cam["camera_center"]),
        prefiltered=False, debug=False)


def hip_forward(cam, sc, bg, flavour=R.WODILATE, use_filter=True, dev="cuda:0", scale_modifier=1.0):
    """Raw backend call (keeps the intermediates).  Returns dict of numpy arrays + the torch `saved`."""
    dev = torch.device(dev)
    rs = settings(cam, bg, dev, scale_modifier)
    t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
    m, s, r, o, c = t(sc["xyz"]), t(sc["scaling"]), t(sc["rotation"]), t(sc["opacity"]).reshape(-1), t(sc["colors"])
    image, radii, pid, pwp, pw, saved = R._backend.forward(rs, flavour, use_filter, m, s, r, o, c)
    torch.cuda.synchronize()
    W, H = cam["image_width"], cam["image_height"]
    offs = R.tile_offsets_of(saved, W, H).cpu().numpy().astype(np.uint32)
    I = int(offs[-1])
    out = dict(image=image.cpu().numpy(), radii=radii.cpu().numpy(),
               rec=saved["geom"].cpu().numpy().reshape(-1, 16)[:, :12], tile_offsets=offs,
               point_list=saved["plist"].cpu().numpy().astype(np.uint32)[:I], I=I,
               final_T=saved["final_T"].cpu().numpy(), n_contrib=saved["n_contrib"].cpu().numpy())
    if pid is not None:
        out.update(point_id_pixel=pid.cpu().numpy(), point_weight_pixel=pwp.cpu().numpy(),
                   point_weight=pw.cpu().numpy())
    out["_torch"] = (rs, flavour, use_filter, m, s, r, saved)
    return out


def hip_backward(hf, dL):
    rs, flavour, use_filter, m, s, r, saved = hf["_torch"]
    g = torch.tensor(np.ascontiguousarray(dL, np.float32), device=m.device)
    g_m3, g_m2, g_c, g_o, g_s, g_r = R._backend.backward(rs, flavour, use_filter, m, s, r, saved, g)
    torch.cuda.synchronize()
    return dict(conic=R._backend.last_conic_grad.cpu().numpy(), means3D=g_m3.cpu().numpy(), means2D=g_m2.cpu().numpy(), colors=g_c.cpu().numpy(),
                opacities=g_o.cpu().numpy().reshape(-1, 1), scales=g_s.cpu().numpy(), rotations=g_r.cpu().numpy())


def hip_project_backward(hf, g_mean2d, g_conic):
    rs, flavour, use_filter, m, s, r, saved = hf["_torch"]
    t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=m.device)
    g3, gs, gr = R._backend.project_backward(rs, flavour, use_filter, m, s, r, saved["radii"], t(g_mean2d), t(g_conic))
    torch.cuda.synchronize()
    return dict(means3D=g3.cpu().numpy(), scales=gs.cpu().numpy(), rotations=gr.cpu().numpy())


def oracle_forward(oracle, cam, sc, bg, flavour=R.WODILATE, use_filter=True, scale_modifier=1.0):
    tfx, tfy = cam_tan(cam)
    fm = flavour.filter_mode if use_filter else _lib.FILTER_NONE
    v = oracle.make_view(cam["image_width"], cam["image_height"], tfx, tfy, cam["world_view_transform"],
                         cam["full_proj_transform"], bg, scale_modifier=scale_modifier, filter_mode=fm,
                         ndc_cull=flavour.ndc_cull)
    f = oracle.forward(v, sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["colors"],
                       extras=bool(flavour.extras))
    return v, f


def compare_forward(hf, of):
    """-> dict of mismatch statistics (0 / 0.0 everywhere = bit-exact)."""
    vis = of["radii"] > 0
    st = {}
    st["radii_mismatch"] = int((hf["radii"] != of["radii"]).sum())
    st["I_hip"], st["I_oracle"] = hf["I"], of["I"]
    rec_h, rec_o = hf["rec"][: len(vis)][vis], of["rec"][vis]
    st["rec_bits_mismatch"] = int((rec_h.view(np.uint32) != rec_o.view(np.uint32)).sum()) if vis.any() else 0
    st["rec_max_abs"] = float(np.abs(rec_h[:, :10] - rec_o[:, :10]).max()) if vis.any() else 0.0
    st["offsets_mismatch"] = int((hf["tile_offsets"] != of["tile_offsets"]).sum())
    same_len = len(hf["point_list"]) == len(of["point_list"])
    st["list_mismatch"] = int((hf["point_list"] != of["point_list"]).sum()) if same_len else -1
    st["n_contrib_mismatch"] = int((hf["n_contrib"] != of["n_contrib"]).sum())
    for k in ("image", "final_T"):
        st[k + "_bits_mismatch"] = int((hf[k].view(np.uint32) != of[k].view(np.uint32)).sum())
        st[k + "_max_abs"] = float(np.abs(hf[k] - of[k]).max())
    if "point_id_pixel" in hf and of.get("point_id_pixel") is not None:
        st["pid_mismatch"] = int((hf["point_id_pixel"] != of["point_id_pixel"]).sum())
        st["pwp_max_abs"] = float(np.abs(hf["point_weight_pixel"] - of["point_weight_pixel"]).max())
        st["pw_max_abs"] = float(np.abs(hf["point_weight"] - of["point_weight"]).max()) if len(of["point_weight"]) else 0.0
    return st

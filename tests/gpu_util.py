"""Helpers for the -m gpu parity tests: run the HIP path (through log_amd's C-ABI binding) and the CPU
oracle on the same inputs."""
import math

import numpy as np
import torch

from log_amd import _lib, rasterizer as R
from util import cam_tan

R._debug_keep = True   # keep dL/dconic of the last backward for the comparisons below


def settings(cam, bg, dev, scale_modifier=1.0):
    tfx, tfy = cam_tan(cam)
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=dev)
    return R.GaussianRasterizationSettings(
        image_height=cam["image_height"], image_width=cam["image_width"], tanfovx=tfx, tanfovy=tfy, bg=t(bg),
        scale_modifier=scale_modifier, viewmatrix=t(cam["world_view_transform"]),
        projmatrix=t(cam["full_proj_transform"]), sh_degree=0, campos=t(cam["camera_center"]),
        prefiltered=False, debug=False)


def hip_forward(cam, sc, bg, flavour=R.WODILATE, use_filter=True, dev="cuda:0", scale_modifier=1.0, scratch_floats=0):
    """Raw backend call (keeps the intermediates).  Returns dict of numpy arrays + the torch `saved`.
    scratch_floats=11: have the forward prepare the backward's accumulators, as the autograd path does."""
    dev = torch.device(dev)
    rs = settings(cam, bg, dev, scale_modifier)
    t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
    m, s, r, o, c = t(sc["xyz"]), t(sc["scaling"]), t(sc["rotation"]), t(sc["opacity"]).reshape(-1), t(sc["colors"])
    image, radii, pid, pwp, pw, saved = R._backend.forward(rs, flavour, use_filter, m, s, r, o, c,
                                                           scratch_floats=scratch_floats)
    torch.cuda.synchronize()
    W, H = cam["image_width"], cam["image_height"]
    offs = R.tile_offsets_of(saved, W, H).cpu().numpy().astype(np.uint32)
    I = int(offs[-1])
    out = dict(image=image.cpu().numpy(), radii=radii.cpu().numpy(),
               rec=saved["geom"][:16 * len(radii)].cpu().numpy().reshape(-1, 16)[:, :12], tile_offsets=offs,
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
    conic = R._backend.last_conic_grad.clone()
    if saved.get("point_weight") is not None:
        # rows of Gaussians that contributed to no pixel are neither cleared nor read on large inputs (their dL/dconic
        # is zero by construction; lograst.h: LOGRAST_BWD_CONIC_TOUCHED_ONLY)
        conic[saved["point_weight"] == 0] = 0
    return dict(conic=conic.cpu().numpy(), means3D=g_m3.cpu().numpy(), means2D=g_m2.cpu().numpy(), colors=g_c.cpu().numpy(),
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
    f["_view"] = v
    return v, f


def _pairs(offsets, plist):
    """(tile << 32 | id) per list entry, in list order."""
    lens = np.diff(offsets.astype(np.int64))
    tile = np.repeat(np.arange(len(lens), dtype=np.uint64), lens)
    return (tile << np.uint64(32)) | plist.astype(np.uint64)


def _compare_culled_lists(hf, of):
    """Support cull on (log_amd/csrc/project.hip): every HIP tile list must be the oracle's list of that tile
    minus entries that cannot pass the alpha floor anywhere in the tile (oracle.instance_support), in the same
    order; n_contrib then counts positions in the shorter list, so it is compared through the identity of the
    last contributor.  Same keys as the exact comparison, counting violations."""
    from oracle import oracle
    st = {}
    ph, po = _pairs(hf["tile_offsets"], hf["point_list"]), _pairs(of["tile_offsets"], of["point_list"])
    # the oracle's pairs are unique and, per tile, in list order; position of every HIP pair in the oracle list
    srt = np.argsort(po, kind="stable")
    idx = np.searchsorted(po[srt], ph)
    idx = np.minimum(idx, len(po) - 1) if len(po) else idx
    found = (po[srt][idx] == ph) if len(po) else np.zeros(len(ph), bool)
    pos = srt[idx] if len(po) else idx                       # index into the oracle's point_list
    not_subset = int((~found).sum())
    # order: within a tile, oracle positions must be strictly increasing (pairs carry the tile in the high bits, and
    # oracle positions grow with the tile, so one global check covers tile boundaries too)
    disorder = int((np.diff(pos.astype(np.int64)) <= 0).sum()) if len(pos) > 1 else 0
    kept = np.zeros(len(po), bool)
    kept[pos[found]] = True
    support = oracle.instance_support(of["_view"], of).astype(bool)
    wrongly_dropped = int((support & ~kept).sum())
    st["culled_instances"] = int((~kept).sum())
    st["offsets_mismatch"] = not_subset + (0 if hf["tile_offsets"][-1] == len(hf["point_list"]) else 1)
    st["list_mismatch"] = not_subset + disorder + wrongly_dropped
    # last contributor per pixel: same Gaussian (or none) on both sides
    H, W = of["n_contrib"].shape
    gx = (W + 15) // 16
    ys, xs = np.mgrid[0:H, 0:W]
    tile = (ys // 16) * gx + xs // 16

    def last_id(f):
        n = f["n_contrib"].astype(np.int64)
        at = f["tile_offsets"].astype(np.int64)[tile] + n - 1
        return np.where(n > 0, f["point_list"][np.clip(at, 0, max(len(f["point_list"]) - 1, 0))].astype(np.int64), -1)

    st["n_contrib_mismatch"] = int((last_id(hf) != last_id(of)).sum()) if len(of["point_list"]) and len(hf["point_list"]) \
        else int(((hf["n_contrib"] > 0) != (of["n_contrib"] > 0)).sum())
    return st


def compare_forward(hf, of):
    """-> dict of mismatch statistics (0 / 0.0 everywhere = bit-exact)."""
    vis = of["radii"] > 0
    st = {}
    st["radii_mismatch"] = int((hf["radii"] != of["radii"]).sum())
    st["I_hip"], st["I_oracle"] = hf["I"], of["I"]
    rec_h, rec_o = hf["rec"][: len(vis)][vis], of["rec"][vis]
    st["rec_bits_mismatch"] = int((rec_h.view(np.uint32) != rec_o.view(np.uint32)).sum()) if vis.any() else 0
    st["rec_max_abs"] = float(np.abs(rec_h[:, :10] - rec_o[:, :10]).max()) if vis.any() else 0.0
    if hf["I"] == of["I"]:
        # same instance count: the lists must be the oracle's lists, entry for entry
        st["offsets_mismatch"] = int((hf["tile_offsets"] != of["tile_offsets"]).sum())
        st["list_mismatch"] = int((hf["point_list"] != of["point_list"]).sum())
        st["n_contrib_mismatch"] = int((hf["n_contrib"] != of["n_contrib"]).sum())
        st["culled_instances"] = 0
    else:
        st.update(_compare_culled_lists(hf, of))
    for k in ("image", "final_T"):
        st[k + "_bits_mismatch"] = int((hf[k].view(np.uint32) != of[k].view(np.uint32)).sum())
        st[k + "_max_abs"] = float(np.abs(hf[k] - of[k]).max())
    if "point_id_pixel" in hf and of.get("point_id_pixel") is not None:
        st["pid_mismatch"] = int((hf["point_id_pixel"] != of["point_id_pixel"]).sum())
        st["pwp_max_abs"] = float(np.abs(hf["point_weight_pixel"] - of["point_weight_pixel"]).max())
        st["pw_max_abs"] = float(np.abs(hf["point_weight"] - of["point_weight"]).max()) if len(of["point_weight"]) else 0.0
    return st

"""Shared helpers for the parity tests."""
import math

import numpy as np

from log_amd import scenes


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def cam_tan(cam):
    return math.tan(cam["FoVx"] * 0.5), math.tan(cam["FoVy"] * 0.5)


def small_case(n=150, W=48, H=40, focal=60.0, radius=2.5, seed=0, opacity=None, smax=0.25, view=1, n_views=3):
    cams = scenes.orbit_cameras(n_views, W=W, H=H, focal=focal, radius=radius)
    sc = scenes.random_scene(n, seed=seed, opacity=opacity, smax=smax)
    return cams[view], sc


def oracle_view(oracle, cam, bg=(1.0, 1.0, 1.0), filter_mode=2, ndc_cull=1, scale_modifier=1.0):
    tfx, tfy = cam_tan(cam)
    return oracle.make_view(cam["image_width"], cam["image_height"], tfx, tfy, cam["world_view_transform"],
                            cam["full_proj_transform"], bg, scale_modifier=scale_modifier,
                            filter_mode=filter_mode, ndc_cull=ndc_cull)

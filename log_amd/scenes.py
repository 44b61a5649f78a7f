"""Synthetic workloads for tests and bench.py: random Gaussian scenes + orbit cameras (numpy, host).

The camera dictionaries have exactly the keys/conventions LoG hands to the rasterizer
(/root/reference/LoG/dataset/base.py:20-55 ``prepare_camera``; projection matrix with principal
point from LoG/utils/camera.py:7-28; orbit poses from the ``DemoDataset`` recipe
LoG/dataset/demo.py:25-46 with up='z').  Scene statistics follow SURVEY.md 8(d) / BASELINE.md 3
(the ``apps/check_gui.py:7-17`` generator scaled with N).
"""
import math

import numpy as np


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def projection_matrix(K, H, W, znear, zfar):
    """OpenCV intrinsics -> clip matrix (column-vector form; caller transposes)."""
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = 2 * K[0, 0] / W
    P[0, 1] = 2 * K[0, 1] / W
    P[0, 2] = -1 + 2 * (K[0, 2] / W)
    P[1, 1] = 2 * K[1, 1] / H
    P[1, 2] = -1 + 2 * (K[1, 2] / H)
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    P[3, 2] = 1.0
    return P


def make_camera(R, T, K, W, H, znear=0.1, zfar=100.0):
    """Camera dict in LoG's row-vector convention (x_row @ M)."""
    R = np.asarray(R, np.float64).reshape(3, 3)
    T = np.asarray(T, np.float64).reshape(3, 1)
    K = np.asarray(K, np.float64).reshape(3, 3)
    wv = np.eye(4)
    wv[:3, :3] = R
    wv[:3, 3:] = T
    wv = wv.T
    proj = projection_matrix(K, H, W, znear, zfar).T
    cam = {
        "image_width": int(W), "image_height": int(H),
        "FoVx": focal2fov(K[0, 0], W), "FoVy": focal2fov(K[1, 1], H),
        "K": K.astype(np.float32),
        "R": R.astype(np.float32), "T": T.astype(np.float32),
        "camera_center": (-R.T @ T).reshape(3).astype(np.float32),
        "world_view_transform": wv.astype(np.float32),
        "full_proj_transform": (wv @ proj).astype(np.float32),
        "znear": znear, "zfar": zfar,
    }
    return cam


def orbit_cameras(n_views=8, radius=3.0, center=(0.0, 0.0, 0.0), W=1920, H=1080, focal=2139.0,
                  start_deg=0.0, end_deg=315.0, znear=0.1, zfar=100.0):
    """Cameras on a circle in the z=0 plane looking at ``center`` (up = +z)."""
    thetas = np.deg2rad(np.linspace(start_deg, end_deg, n_views))
    K = np.array([[focal, 0, W / 2.0], [0, focal, H / 2.0], [0, 0, 1.0]])
    cams = []
    c0 = np.asarray(center, np.float64).reshape(3)
    for th in thetas:
        st, ct = math.sin(th), math.cos(th)
        R = np.array([[-st, ct, 0.0], [0.0, 0.0, -1.0], [-ct, -st, 0.0]])
        cpos = np.array([radius * ct, radius * st, 0.0]) + c0
        T = -R @ cpos.reshape(3, 1)
        cams.append(make_camera(R, T, K, W, H, znear, zfar))
    return cams


def random_scene(n, seed=0, opacity=0.999, smax=None, extent=1.0):
    """xyz in a cube of side ``extent``, scales U(0, smax) with smax = 0.5 n^(-1/3), unit quaternions."""
    rng = np.random.default_rng(seed)
    if smax is None:
        smax = 0.5 * float(n) ** (-1.0 / 3.0)
    xyz = ((rng.random((n, 3), dtype=np.float32) - 0.5) * extent).astype(np.float32)
    scales = (rng.random((n, 3), dtype=np.float32) * smax).astype(np.float32)
    rot = rng.random((n, 4), dtype=np.float32) + 1e-3
    rot = (rot / np.linalg.norm(rot, axis=1, keepdims=True)).astype(np.float32)
    if opacity is None:
        opac = rng.random((n, 1), dtype=np.float32)
    else:
        opac = np.full((n, 1), opacity, np.float32)
    colors = rng.random((n, 3), dtype=np.float32)
    return dict(xyz=xyz, scaling=scales, rotation=rot, opacity=opac, colors=colors)

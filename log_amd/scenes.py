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


def trained_like_scene(n, seed=0, extent=1.0, sigma=0.5, max_anisotropy=10.0, opacity_sigma=2.0, smed=None):
    """A scene with the attribute STATISTICS of a trained splat model rather than check_gui's uniform draws (round-4
    verdict, missing #4: U(0, s_max) scales are needles and pancakes by construction, and opacity 0.999 hides 93 % of the
    Gaussians): log-normal scales (per-Gaussian size exp(sigma z) around `smed`, per-axis factors exp(sigma z_k)) with the
    anisotropy max/min clamped to `max_anisotropy`, opacity = sigmoid(opacity_sigma z) (bimodal towards 0 and 1), uniformly
    random orientations.  smed defaults to 0.25 n^(-1/3): the MEAN scale of random_scene, so the screen coverage is alike."""
    rng = np.random.default_rng(seed)
    if smed is None:
        smed = 0.25 * float(n) ** (-1.0 / 3.0)
    xyz = ((rng.random((n, 3), dtype=np.float32) - 0.5) * extent).astype(np.float32)
    size = np.exp(sigma * rng.standard_normal((n, 1), dtype=np.float32))
    axes = np.exp(sigma * rng.standard_normal((n, 3), dtype=np.float32))
    scales = (smed * size * axes).astype(np.float32)
    scales = np.maximum(scales, scales.max(axis=1, keepdims=True) / np.float32(max_anisotropy)).astype(np.float32)
    rot = rng.standard_normal((n, 4), dtype=np.float32)
    rot = (rot / np.maximum(np.linalg.norm(rot, axis=1, keepdims=True), 1e-12)).astype(np.float32)
    opac = (1.0 / (1.0 + np.exp(-opacity_sigma * rng.standard_normal((n, 1), dtype=np.float32)))).astype(np.float32)
    colors = rng.random((n, 3), dtype=np.float32)
    return dict(xyz=xyz, scaling=scales, rotation=rot, opacity=opac, colors=colors)


# ---- synthetic level-of-detail trees (BASELINE.json configs[2]: "LoG LoD tree ... level selection on") -----------
# The buffer layout TensorTree keeps (/root/reference/LoG/model/tensor_tree.py:57-90), built level by level the way
# `split` appends nodes and children, plus holes (tree entries set to -1) the way `remove` leaves them.  Used by the
# C3 leg of bench.py and by the tests where the golden trees (grown by the reference's own class) are too small.
def synth_tree(n_roots, n_levels, max_child, split_prob=0.75, hole_prob=0.03, seed=0, extent=1.0, root_scale=0.05):
    rng = np.random.default_rng(seed)
    xyz = ((rng.random((n_roots, 3)) - 0.5) * extent).astype(np.float32)
    scaling = np.log(rng.random((n_roots, 3)) * root_scale + 0.2 * root_scale).astype(np.float32)
    rotation = rng.standard_normal((n_roots, 4)).astype(np.float32)
    node_index = np.full(n_roots, -1, np.int32)
    depth = np.zeros(n_roots, np.int8)
    rows = []
    num_nodes = 0
    for level in range(n_levels):
        cand = np.nonzero((node_index == -1) & (depth == level))[0]
        parent = cand[rng.random(cand.shape[0]) < split_prob]
        ns = parent.shape[0]
        if ns == 0:
            break
        P = node_index.shape[0]
        node_index[parent] = num_nodes + np.arange(ns, dtype=np.int32)
        child = (P + np.arange(ns * max_child, dtype=np.int32)).reshape(ns, max_child)
        rows.append(child)
        num_nodes += ns
        rep = np.repeat(parent, max_child)
        sig = np.exp(scaling[rep]).max(axis=1, keepdims=True)
        xyz = np.concatenate([xyz, xyz[rep] + (rng.standard_normal((rep.shape[0], 3)) * sig).astype(np.float32)])
        scaling = np.concatenate([scaling, (scaling[rep] - math.log(1.6)
                                            + 0.2 * rng.standard_normal((rep.shape[0], 3))).astype(np.float32)])
        rotation = np.concatenate([rotation, rotation[rep] + 0.3 * rng.standard_normal((rep.shape[0], 4)).astype(np.float32)])
        node_index = np.concatenate([node_index, np.full(rep.shape[0], -1, np.int32)])
        depth = np.concatenate([depth, np.full(rep.shape[0], level + 1, np.int8)])
    tree = np.concatenate(rows) if rows else np.zeros((0, max_child), np.int32)
    if hole_prob > 0 and tree.size:
        holes = (rng.random(tree.shape) < hole_prob) & (node_index[tree] == -1)     # only leaves are ever removed
        tree = np.where(holes, -1, tree).astype(np.int32)
    return dict(node_index=node_index, tree=np.ascontiguousarray(tree), depth=depth,
                xyz=np.ascontiguousarray(xyz, np.float32), scaling=np.ascontiguousarray(scaling, np.float32),
                rotation=np.ascontiguousarray(rotation, np.float32), root_index=np.arange(n_roots, dtype=np.int64))

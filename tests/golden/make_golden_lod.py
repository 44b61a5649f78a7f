"""Generates tests/golden/lod_*.npz by RUNNING the reference's own level-of-detail code (build container only:
needs /root/reference; the fixtures are committed and travel to the GPU box).

What is pinned: TensorTree.initialize / split / remove build the tree buffers, TensorTree.traverse +
_query_tree_torch (LoG/model/tensor_tree.py:131-185) select the points, Gaussian.compute_radius
(LoG/model/level_of_gaussian.py:65-88) gathers and activates (torch.exp, F.normalize) -- all unmodified
reference code.  The only stand-in is compute_radius_module (the CUDA extension, not buildable here), for
which the oracle's A0 is used (itself pinned against LoG/model/geometry.py by make_golden.py).

    python tests/golden/make_golden_lod.py
"""
import math
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("LOG_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (REF, ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def reference_env():
    """Stubs for host-only deps + the LoG.cuda drop-in backed by the oracle (CPU)."""
    from log_amd import rasterizer as R
    from log_amd.compute_radius import compute_radius_module
    import oracle_backend
    from oracle_backend import OracleBackend
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    drop = types.ModuleType("LoG.cuda.compute_radius")
    drop.compute_radius_module = compute_radius_module
    sys.modules["LoG.cuda.compute_radius"] = drop
    oracle_backend.install(OracleBackend())


def build_case(seed, n_roots, n_levels, max_child, split_prob=0.7, remove_prob=0.04, extent=1.0):
    """A tree grown with the reference's TensorTree, parameters of children derived from their parent."""
    from LoG.model.tensor_tree import TensorTree
    g = torch.Generator().manual_seed(seed)
    tree = TensorTree(max_child=max_child, max_level=30)
    xyz = (torch.rand(n_roots, 3, generator=g) - 0.5) * extent
    scaling = torch.log(torch.rand(n_roots, 3, generator=g) * 0.06 + 0.01)
    rotation = torch.randn(n_roots, 4, generator=g)
    tree.initialize(xyz)
    for level in range(n_levels):
        cand = torch.where((tree.node_index == -1) & (tree.depth == level))[0]
        parent = cand[torch.rand(cand.shape[0], generator=g) < split_prob]
        if parent.numel() == 0:
            break
        tree.split(parent)
        rep = parent[:, None].repeat(1, max_child).reshape(-1)
        sig = torch.exp(scaling[rep]).max(dim=-1, keepdim=True).values
        xyz = torch.cat([xyz, xyz[rep] + torch.randn(rep.shape[0], 3, generator=g) * sig])
        scaling = torch.cat([scaling, scaling[rep] - math.log(1.6) + 0.2 * torch.randn(rep.shape[0], 3, generator=g)])
        rotation = torch.cat([rotation, rotation[rep] + 0.3 * torch.randn(rep.shape[0], 4, generator=g)])
    # holes: remove some non-root leaves (tensor_tree.py:92-119); the caller drops the same rows of the parameters
    cand = torch.where((tree.node_index == -1) & (tree.index_parent > -1))[0]
    rem = cand[torch.rand(cand.shape[0], generator=g) < remove_prob]
    keep = torch.ones(tree.num_points, dtype=torch.bool)
    keep[rem] = False
    tree.remove(rem)
    xyz, scaling, rotation = xyz[keep], scaling[keep], rotation[keep]
    assert xyz.shape[0] == tree.num_points
    return tree, xyz.contiguous(), scaling.contiguous(), rotation.contiguous()


def camera_and_rasterizer(W, H, focal, theta=30.0, radius=2.5):
    from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer
    from log_amd import scenes
    cam = scenes.orbit_cameras(8, W=W, H=H, focal=focal, radius=radius)[int(theta // 45) % 8]
    tfx, tfy = math.tan(cam["FoVx"] * 0.5), math.tan(cam["FoVy"] * 0.5)
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=tfx, tanfovy=tfy, bg=torch.zeros(3), scale_modifier=1.0,
        viewmatrix=torch.tensor(cam["world_view_transform"]), projmatrix=torch.tensor(cam["full_proj_transform"]),
        sh_degree=0, campos=torch.tensor(cam["camera_center"]), prefiltered=False, debug=False)
    return cam, GaussianRasterizer(raster_settings=rs)


def reference_traverse(tree, xyz, scaling, rotation, root_index, rast, min_px, max_depth):
    from LoG.model.level_of_gaussian import Gaussian
    g = Gaussian()
    g.xyz, g.scaling, g.rotation = xyz, scaling, rotation
    tree.min_resolution_pixel = min_px
    return tree.traverse(g, root_index, rast, max_depth=max_depth)


CASES = [dict(name="a", seed=3, n_roots=300, n_levels=4, max_child=4, W=640, H=480, focal=600.0, theta=45.0),
         dict(name="b", seed=4, n_roots=1500, n_levels=3, max_child=2, W=400, H=400, focal=445.0, theta=180.0)]
QUERIES = [(3.0, 1000), (3.0, 2), (3.0, 0), (3.0, 1), (8.0, 1000), (0.5, 1000), (1e9, 1000)]


def main():
    reference_env()
    for c in CASES:
        tree, xyz, scaling, rotation = build_case(c["seed"], c["n_roots"], c["n_levels"], c["max_child"])
        cam, rast = camera_and_rasterizer(c["W"], c["H"], c["focal"], c["theta"])
        g = torch.Generator().manual_seed(c["seed"] + 100)
        roots = tree.root_index.long()
        roots = roots[torch.rand(roots.shape[0], generator=g) < 0.9]       # LoG.prepare hands over a subset
        out = {"node_index": tree.node_index.numpy(), "tree": tree.tree.numpy(), "depth": tree.depth.numpy(),
               "xyz": xyz.numpy(), "scaling": scaling.numpy(), "rotation": rotation.numpy(),
               "root_index": roots.numpy(), "max_level": np.int32(tree.max_level),
               "viewmatrix": np.asarray(cam["world_view_transform"], np.float32),
               "projmatrix": np.asarray(cam["full_proj_transform"], np.float32),
               "wh": np.array([c["W"], c["H"]], np.int32),
               "tanfov": np.array([rast.raster_settings.tanfovx, rast.raster_settings.tanfovy], np.float64),
               "queries": np.array(QUERIES, np.float64)}
        for qi, (min_px, max_depth) in enumerate(QUERIES):
            idx = reference_traverse(tree, xyz, scaling, rotation, roots, rast, min_px, int(max_depth))
            out[f"index_{qi}"] = idx.numpy().astype(np.int64)
            print(c["name"], "min_px", min_px, "max_depth", max_depth, "->", idx.shape[0], "of", tree.num_points,
                  "points; mean depth", float(tree.depth[idx].float().mean()))
        np.savez_compressed(os.path.join(HERE, f"lod_{c['name']}.npz"), **out)


if __name__ == "__main__":
    main()

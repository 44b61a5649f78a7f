"""N3 on the GPU: lograst_lod_traverse (through the drop-in, log_amd/lod.py) against the index lists of the
reference's own TensorTree.traverse (tests/golden/lod_*.npz) and, at larger sizes, against the oracle -- integer
output, compared exactly, order included."""
import glob
import math
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "lod_*.npz")))


def _objects(node_index, tree, xyz, scaling, rotation, W, H, tfx, tfy, viewmatrix, projmatrix, max_level=30):
    from log_amd.rasterizer import GaussianRasterizationSettings
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tr = types.SimpleNamespace(node_index=t(node_index), tree=t(tree), max_level=max_level, min_resolution_pixel=3)
    act = types.SimpleNamespace(scaling_activation=torch.exp, rotation_activation=torch.nn.functional.normalize)
    model = types.SimpleNamespace(xyz=t(xyz), scaling=t(scaling), rotation=t(rotation), activation=act)
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=tfx, tanfovy=tfy, bg=torch.zeros(3, device=dev), scale_modifier=1.0,
        viewmatrix=t(np.asarray(viewmatrix, np.float32)), projmatrix=t(np.asarray(projmatrix, np.float32)),
        sh_degree=0, campos=torch.zeros(3, device=dev), prefiltered=False, debug=False)
    return tr, model, types.SimpleNamespace(raster_settings=rs)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_traverse_matches_reference_lists(path):
    from log_amd import lod
    g = np.load(path)
    W, H = (int(v) for v in g["wh"])
    tree, model, cam = _objects(g["node_index"], g["tree"], g["xyz"], g["scaling"], g["rotation"], W, H,
                                float(g["tanfov"][0]), float(g["tanfov"][1]), g["viewmatrix"], g["projmatrix"],
                                int(g["max_level"]))
    roots = torch.from_numpy(g["root_index"]).cuda()
    for qi, (min_px, max_depth) in enumerate(g["queries"]):
        tree.min_resolution_pixel = float(min_px)
        got = lod.traverse(tree, model, roots, cam, max_depth=int(max_depth))
        assert got.dtype == torch.int64 and got.is_cuda
        np.testing.assert_array_equal(got.cpu().numpy(), g[f"index_{qi}"],
                                      err_msg=f"query {qi}: min_px {min_px} max_depth {max_depth}")


@pytest.mark.parametrize("n_roots,levels,max_child,min_px", [(20000, 5, 4, 3.0), (3000, 9, 2, 2.0), (50000, 2, 8, 6.0)])
def test_traverse_matches_oracle_on_large_trees(n_roots, levels, max_child, min_px, oracle_mod):
    from log_amd import lod, scenes
    from lod_util import synth_tree
    s = synth_tree(n_roots, levels, max_child, seed=n_roots)
    W, H = 1920, 1080
    cam = scenes.orbit_cameras(8, W=W, H=H)[3]
    tfx, tfy = math.tan(cam["FoVx"] * 0.5), math.tan(cam["FoVy"] * 0.5)
    tree, model, rast = _objects(s["node_index"], s["tree"], s["xyz"], s["scaling"], s["rotation"], W, H, tfx, tfy,
                                 cam["world_view_transform"], cam["full_proj_transform"])
    tree.min_resolution_pixel = min_px
    rng = np.random.default_rng(1)
    roots = s["root_index"][rng.random(n_roots) < 0.8]
    for max_depth in (1000, 1):
        got = lod.traverse(tree, model, torch.from_numpy(roots).cuda(), rast, max_depth=max_depth).cpu().numpy()
        want = oracle_mod.lod_traverse(s["node_index"], s["tree"], s["xyz"], s["scaling"], s["rotation"], roots,
                                       cam["full_proj_transform"], cam["world_view_transform"], W / (2 * tfx),
                                       H / (2 * tfy), tfx, tfy, min_px, 30, max_depth)
        np.testing.assert_array_equal(got, want)
        assert len(np.unique(got)) == len(got)                      # a point is selected at most once
    assert want.shape[0] > roots.shape[0]                           # the trees really were descended


def test_traverse_edge_cases(oracle_mod):
    from log_amd import lod, scenes
    from lod_util import synth_tree
    cam = scenes.orbit_cameras(1, W=640, H=480, focal=500.0)[0]
    tfx, tfy = math.tan(cam["FoVx"] * 0.5), math.tan(cam["FoVy"] * 0.5)
    # no nodes at all: every root is a leaf and is kept, in order
    s = synth_tree(1000, 0, 4, seed=2)
    tree, model, rast = _objects(s["node_index"], s["tree"], s["xyz"], s["scaling"], s["rotation"], 640, 480, tfx, tfy,
                                 cam["world_view_transform"], cam["full_proj_transform"])
    roots = torch.arange(999, -1, -3, device="cuda")
    got = lod.traverse(tree, model, roots, rast)
    np.testing.assert_array_equal(got.cpu().numpy(), roots.cpu().numpy())
    # no roots
    assert lod.traverse(tree, model, torch.zeros(0, dtype=torch.int64, device="cuda"), rast).numel() == 0
    # int32 root indices (TensorTree.root_index is int32 before LoG's .long()) and a non-contiguous selection
    s = synth_tree(500, 3, 4, seed=3)
    tree, model, rast = _objects(s["node_index"], s["tree"], s["xyz"], s["scaling"], s["rotation"], 640, 480, tfx, tfy,
                                 cam["world_view_transform"], cam["full_proj_transform"])
    roots32 = torch.arange(0, 500, dtype=torch.int32, device="cuda")[::2]
    got = lod.traverse(tree, model, roots32, rast).cpu().numpy()
    want = oracle_mod.lod_traverse(s["node_index"], s["tree"], s["xyz"], s["scaling"], s["rotation"],
                                   np.arange(0, 500, 2), cam["full_proj_transform"], cam["world_view_transform"],
                                   640 / (2 * tfx), 480 / (2 * tfy), tfx, tfy, 3.0, 30, 1000)
    np.testing.assert_array_equal(got, want)


def test_cached_tree_depth_hint_and_stale_hint(oracle_mod):
    """lod.traverse launches only as many levels as the tree is deep (cached on the tree object); a stale cache is
    noticed on the device (frontier left at the hinted depth) and the selection repeated with the full depth."""
    from log_amd import lod, scenes
    from lod_util import synth_tree
    cam = scenes.orbit_cameras(1, W=640, H=480, focal=500.0)[0]
    tfx, tfy = math.tan(cam["FoVx"] * 0.5), math.tan(cam["FoVy"] * 0.5)
    s = synth_tree(2000, 5, 4, seed=11)
    tree, model, rast = _objects(s["node_index"], s["tree"], s["xyz"], s["scaling"], s["rotation"], 640, 480, tfx, tfy,
                                 cam["world_view_transform"], cam["full_proj_transform"])
    tree.depth = torch.from_numpy(s["depth"]).cuda()
    roots = torch.from_numpy(s["root_index"]).cuda()
    want = oracle_mod.lod_traverse(s["node_index"], s["tree"], s["xyz"], s["scaling"], s["rotation"], s["root_index"],
                                   cam["full_proj_transform"], cam["world_view_transform"], 640 / (2 * tfx),
                                   480 / (2 * tfy), tfx, tfy, 3.0, 30, 1000)
    got = lod.traverse(tree, model, roots, rast)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    assert tree._lograst_depth[1] == int(s["depth"].max())
    tree._lograst_depth = (tree._lograst_depth[0], 1)          # pretend the cache is stale
    got = lod.traverse(tree, model, roots, rast)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    # an explicit depth limit below the tree's depth is not a stale hint: the frontier is part of the answer
    want2 = oracle_mod.lod_traverse(s["node_index"], s["tree"], s["xyz"], s["scaling"], s["rotation"], s["root_index"],
                                    cam["full_proj_transform"], cam["world_view_transform"], 640 / (2 * tfx),
                                    480 / (2 * tfy), tfx, tfy, 3.0, 30, 2)
    tree._lograst_depth = None
    np.testing.assert_array_equal(lod.traverse(tree, model, roots, rast, max_depth=2).cpu().numpy(), want2)

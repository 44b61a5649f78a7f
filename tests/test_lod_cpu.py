"""N3: level-of-detail selection.  CPU side: the oracle's restatement of TensorTree.traverse
(oracle/oracle.py:lod_traverse) and the drop-in's host logic (log_amd/lod.py, driven through the oracle test
double) against index lists produced by the reference's own TensorTree / Gaussian.compute_radius
(tests/golden/make_golden_lod.py)."""
import glob
import os
import types

import numpy as np
import pytest
import torch

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "lod_*.npz")))


def _focal(g):
    W, H = (int(v) for v in g["wh"])
    tfx, tfy = (float(v) for v in g["tanfov"])
    return W / (2.0 * tfx), H / (2.0 * tfy), tfx, tfy


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_traverse_matches_reference_lists(path, oracle_mod):
    g = np.load(path)
    fx, fy, tfx, tfy = _focal(g)
    for qi, (min_px, max_depth) in enumerate(g["queries"]):
        got = oracle_mod.lod_traverse(g["node_index"], g["tree"], g["xyz"], g["scaling"], g["rotation"],
                                      g["root_index"], g["projmatrix"], g["viewmatrix"], fx, fy, tfx, tfy,
                                      min_px, int(g["max_level"]), int(max_depth))
        np.testing.assert_array_equal(got, g[f"index_{qi}"], err_msg=f"query {qi}: min_px {min_px} max_depth {max_depth}")


def test_golden_fixtures_present():
    assert len(GOLDEN) >= 2


def _tree_and_model(g):
    """Objects with the attributes log_amd.lod.traverse reads from TensorTree / Gaussian / the rasterizer."""
    from log_amd.rasterizer import GaussianRasterizationSettings
    tree = types.SimpleNamespace(node_index=torch.from_numpy(g["node_index"]), tree=torch.from_numpy(g["tree"]),
                                 max_level=int(g["max_level"]), min_resolution_pixel=3)
    act = types.SimpleNamespace(scaling_activation=torch.exp, rotation_activation=torch.nn.functional.normalize)
    model = types.SimpleNamespace(xyz=torch.from_numpy(g["xyz"]), scaling=torch.from_numpy(g["scaling"]),
                                  rotation=torch.from_numpy(g["rotation"]), activation=act)
    W, H = (int(v) for v in g["wh"])
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=float(g["tanfov"][0]), tanfovy=float(g["tanfov"][1]),
        bg=torch.zeros(3), scale_modifier=1.0, viewmatrix=torch.from_numpy(g["viewmatrix"]),
        projmatrix=torch.from_numpy(g["projmatrix"]), sh_degree=0, campos=torch.zeros(3), prefiltered=False,
        debug=False)
    return tree, model, types.SimpleNamespace(raster_settings=rs)


def test_dropin_host_logic_with_test_double(oracle_mod):
    from log_amd import lod, rasterizer as R
    import oracle_backend
    from oracle_backend import OracleBackend
    g = np.load(GOLDEN[-1])
    tree, model, cam = _tree_and_model(g)
    old = oracle_backend.install(OracleBackend())
    try:
        for qi, (min_px, max_depth) in enumerate(g["queries"]):
            tree.min_resolution_pixel = float(min_px)
            got = lod.traverse(tree, model, torch.from_numpy(g["root_index"]), cam, max_depth=int(max_depth))
            assert got.dtype == torch.int64
            np.testing.assert_array_equal(got.numpy(), g[f"index_{qi}"])
        # no roots at all (LoG.prepare can hand over an empty selection)
        got = lod.traverse(tree, model, torch.zeros(0, dtype=torch.int64), cam)
        assert got.numel() == 0
        model.activation.scaling_activation = torch.sigmoid
        with pytest.raises(NotImplementedError):
            lod.traverse(tree, model, torch.from_numpy(g["root_index"]), cam)
    finally:
        oracle_backend.install(None if isinstance(old, R.HipBackend) else old)


def test_product_path_refuses_cpu_tensors():
    from log_amd import lod, _lib
    g = np.load(GOLDEN[-1])
    tree, model, cam = _tree_and_model(g)
    with pytest.raises(_lib.LograstError):
        lod.traverse(tree, model, torch.from_numpy(g["root_index"]), cam)


REF = os.environ.get("LOG_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "LoG")), reason="reference tree not present")
@pytest.mark.parametrize("seed", range(8))
def test_oracle_and_dropin_follow_reference_on_random_trees(seed, oracle_mod):
    """Fresh trees grown and pruned by the reference's TensorTree (different arity, depth, removal rate per seed), a
    random camera and threshold: the reference's traverse, the oracle and the drop-in (through the test double)
    return the same list."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden_lod as G
    import oracle_backend
    from log_amd import lod, rasterizer as R
    rng = np.random.default_rng(seed)
    G.reference_env()                                     # installs the oracle test double + LoG.cuda stand-in
    try:
        mc = int(rng.choice([2, 3, 4, 8]))
        tree, xyz, scaling, rotation = G.build_case(100 + seed, int(rng.integers(50, 400)), int(rng.integers(1, 5)), mc,
                                                    split_prob=float(rng.uniform(0.3, 0.95)),
                                                    remove_prob=float(rng.uniform(0.0, 0.3)))
        cam, rast = G.camera_and_rasterizer(320, 240, float(rng.uniform(150, 500)), theta=float(rng.integers(0, 8) * 45))
        roots = tree.root_index.long()
        roots = roots[torch.from_numpy(rng.random(roots.shape[0]) < 0.85)]
        for _ in range(3):
            min_px = float(rng.choice([0.5, 2.0, 3.0, 6.0, 20.0]))
            max_depth = int(rng.choice([0, 1, 2, 3, 1000]))
            want = G.reference_traverse(tree, xyz, scaling, rotation, roots, rast, min_px, max_depth).numpy()
            rs = rast.raster_settings
            fx, fy = rs.image_width / (2 * rs.tanfovx), rs.image_height / (2 * rs.tanfovy)
            got = oracle_mod.lod_traverse(tree.node_index.numpy(), tree.tree.numpy(), xyz.numpy(), scaling.numpy(),
                                          rotation.numpy(), roots.numpy(), rs.projmatrix.numpy(), rs.viewmatrix.numpy(),
                                          fx, fy, rs.tanfovx, rs.tanfovy, min_px, tree.max_level, max_depth)
            np.testing.assert_array_equal(got, want)
            g = types.SimpleNamespace(xyz=xyz, scaling=scaling, rotation=rotation,
                                      activation=types.SimpleNamespace(scaling_activation=torch.exp,
                                                                       rotation_activation=torch.nn.functional.normalize))
            tree.min_resolution_pixel = min_px
            np.testing.assert_array_equal(lod.traverse(tree, g, roots, rast, max_depth=max_depth).numpy(), want)
    finally:
        oracle_backend.install(None)
        for k in ("LoG.cuda.compute_radius",):
            sys.modules.pop(k, None)

"""The performance knobs of liblograst (lograst_set_knob; log_amd/tune.py) change launch shapes, never results: every
knob swept over several values on a tree-ordered, heavy-tailed view -- outputs, tile lists and fork maps bit for bit,
gradients to summation-order noise -- and the calibration itself run end to end on a small workload."""
import json
import os

import numpy as np
import pytest
import torch

from util import rel_l2

pytestmark = pytest.mark.gpu

SWEEP = {
    "LOGRAST_HELPER_MIN_N": (0, 4_000_000, 2_000_000_000),
    "LOGRAST_DEFER_TILES": (4, 16, 100),
    "LOGRAST_MID_COOP": (0, 1, 16, 64),
    "LOGRAST_MID_RANK": (0, 1),
    "LOGRAST_PBWD_LIST": (0, 1, 2),
    "LOGRAST_LAZY_SORT": (0, 1),
    "LOGRAST_HIT_MASKS": (0, 1),
    "LOGRAST_HUGE_CHUNK": (256, 512, 2048),
    "LOGRAST_BATCH_PLANES": (1, 2, 4),
    "LOGRAST_BATCH_SLOTS": (64, 256, 1024),
    "LOGRAST_SEPARATE_ZERO": (0, 1),
    "LOGRAST_FILL_XCD_ORDER": (0, 1),
    "LOGRAST_FILL_NT": (0, 1),
    "LOGRAST_XCD_MODE": (0, 1, 2, 3),
    "LOGRAST_PROJECT_BLOCKS": (64, 512, 4096),
    "LOGRAST_BWD_ROWS": (0, 1, 2),
    "LOGRAST_FWD_ROWS": (0, 1, 2),
    "LOGRAST_BWD_BLOCK_TEST": (0, 1),
    "LOGRAST_FWD_BLOCK_TEST": (0, 1),
    "LOGRAST_BAND_SPARSE": (0, 1),
    "LOGRAST_FILL_PER_THREAD": (1, 2, 4),
    "LOGRAST_FILL_STAGED": (0, 1, 2, 3),
}


def _tree_view():
    """The level-of-detail selection of a small tree: siblings in neighbouring rows, a heavy tail of large rects."""
    import types
    from diff_gaussian_rasterization_wodilate import GaussianRasterizer
    import gpu_util as G
    from log_amd import lod, scenes
    dev = torch.device("cuda:0")
    W, H = 1280, 720
    tr = scenes.synth_tree(12000, 6, 4, split_prob=0.5, hole_prob=0.02, seed=2, root_scale=0.05)
    cam = scenes.orbit_cameras(8, W=W, H=H, focal=1400.0)[2]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rast = GaussianRasterizer(raster_settings=G.settings(cam, (1.0, 1.0, 1.0), dev))
    tree = types.SimpleNamespace(node_index=t(tr["node_index"]), tree=t(tr["tree"]), max_level=30, min_resolution_pixel=3.0)
    act = types.SimpleNamespace(scaling_activation=torch.exp, rotation_activation=torch.nn.functional.normalize)
    model = types.SimpleNamespace(xyz=t(tr["xyz"]), scaling=t(tr["scaling"]), rotation=t(tr["rotation"]), activation=act)
    sel = lod.traverse(tree, model, t(tr["root_index"]), rast).cpu().numpy()
    rng = np.random.default_rng(5)
    q = tr["rotation"][sel]
    sc = dict(xyz=tr["xyz"][sel], scaling=np.exp(tr["scaling"][sel]).astype(np.float32),
              rotation=(q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)).astype(np.float32),
              opacity=(1.0 / (1.0 + np.exp(-(rng.standard_normal((sel.shape[0], 1)) + 1.0)))).astype(np.float32),
              colors=rng.random((sel.shape[0], 3), dtype=np.float32))
    return cam, sc


def test_every_knob_leaves_every_output_bit_identical():
    import gpu_util as G
    from log_amd import tune
    cam, sc = _tree_view()
    dL = np.random.default_rng(3).random((3, cam["image_height"], cam["image_width"]), dtype=np.float32)
    tune.reset_knobs()
    names = {k["name"] for k in tune.knobs()}
    assert set(SWEEP) == names, names ^ set(SWEEP)
    ref = G.hip_forward(cam, sc, (1.0, 1.0, 1.0), scratch_floats=16)
    assert int((ref["radii"] > 16).sum()) > 100 and ref["I"] > 2 * len(sc["xyz"])
    g_ref = G.hip_backward(ref, dL)
    try:
        for name, values in SWEEP.items():
            for val in values:
                tune.set_knob(name, val)
                assert tune.get_knob(name) == val
                hf = G.hip_forward(cam, sc, (1.0, 1.0, 1.0), scratch_floats=16)
                for k in ("image", "final_T", "point_weight_pixel", "point_weight"):
                    assert (hf[k].view(np.uint32) == ref[k].view(np.uint32)).all(), (name, val, k)
                for k in ("radii", "tile_offsets", "point_list", "n_contrib", "point_id_pixel"):
                    assert (hf[k] == ref[k]).all(), (name, val, k)
                g = G.hip_backward(hf, dL)
                for k in ("means2D", "conic", "opacities", "colors"):
                    assert rel_l2(g[k], g_ref[k]) < 1e-5, (name, val, k)
            tune.reset_knobs()
        with pytest.raises(Exception, match="unknown knob"):
            tune.set_knob("LOGRAST_NO_SUCH_KNOB", 1)
        with pytest.raises(Exception, match="out of range"):
            tune.set_knob("LOGRAST_BATCH_PLANES", 9)
    finally:
        tune.reset_knobs()


def test_band_views_skip_rectless_gaussians_without_changing_a_bit():
    """A view that owns a band of tile rows (one rank's share of an image split across GPUs) is projected by
    lr_project_band_kernel: Gaussians whose rect misses the band cost neither a record nor a fill record nor their opacity
    / colour, the survivors are compacted into full waves and into per-workgroup slot ranges that the fill and the
    deferred-rect count walk (LOGRAST_BAND_SPARSE).  Against the full-view kernel on the same band, for bands at the top, the middle and the
    bottom of the image, with deferred (huge) rects in play and with the exact two-call forward as well as the
    speculative one: radii, tile lists, image, fork maps bit for bit; records of the Gaussians that have a rect bit for
    bit; gradients to summation-order noise."""
    import gpu_util as G
    from log_amd import rasterizer as R, tune
    cam, sc = _tree_view()
    H = cam["image_height"]
    gy = (H + 15) // 16
    dL = np.random.default_rng(4).random((3, H, cam["image_width"]), dtype=np.float32)
    tune.reset_knobs()
    try:
        for rows in ((0, 7), (gy // 2 - 3, gy // 2 + 4), (gy - 6, gy)):
            for speculative in (False, True):
                prev = R.set_speculative(speculative)
                try:
                    res = {}
                    for sparse in (0, 1):
                        tune.set_knob("LOGRAST_BAND_SPARSE", sparse)
                        tune.set_knob("LOGRAST_DEFER_TILES", 8)               # more rects through lr_count_huge_kernel
                        with R.tile_rows(*rows):
                            hf = G.hip_forward(cam, sc, (1.0, 1.0, 1.0), scratch_floats=16)
                            res[sparse] = (hf, G.hip_backward(hf, dL))
                finally:
                    R.set_speculative(prev)
                (a, ga), (b, gb) = res[0], res[1]
                vis = a["radii"] > 0
                if rows[0] > 0 and rows[1] < gy:                              # the middle band: a real mix
                    assert 0.01 < vis.mean() < 0.95 and a["I"] > 1000, (rows, vis.mean(), a["I"])
                for k in ("image", "final_T", "point_weight_pixel", "point_weight"):
                    assert (a[k].view(np.uint32) == b[k].view(np.uint32)).all(), (rows, k)
                for k in ("radii", "tile_offsets", "point_list", "n_contrib", "point_id_pixel"):
                    assert (a[k] == b[k]).all(), (rows, k)
                assert (a["rec"][vis].view(np.uint32) == b["rec"][vis].view(np.uint32)).all(), rows
                for k in ga:
                    assert rel_l2(gb[k], ga[k]) < 1e-5, (rows, k)
                    assert np.abs(gb[k][~vis]).max() == 0 or k == "conic", (rows, k)
    finally:
        tune.reset_knobs()


def test_calibration_runs_end_to_end_and_is_reloadable(tmp_path):
    from log_amd import tune
    dev = torch.device("cuda:0")
    views = [tune.synthetic_views(dev, 200_000, 640, 360), tune.synthetic_views(dev, 200_000, 640, 360, heavy_tail=True)]
    path = str(tmp_path / "tune.json")
    try:
        chosen = tune.tune(views=views, device=dev, repeats=2, save=True, path=path, helper=False,
                           candidates={"LOGRAST_DEFER_TILES": (8, 16, 32), "LOGRAST_BWD_ROWS": (0, 1)})
        assert set(chosen) == {"LOGRAST_DEFER_TILES", "LOGRAST_BWD_ROWS"}
        assert chosen["LOGRAST_DEFER_TILES"] in (8, 16, 32) and chosen["LOGRAST_BWD_ROWS"] in (0, 1, 2)
        stored = json.load(open(path))
        assert stored["knobs"] == chosen and "timings_ms" in stored
        tune.reset_knobs()
        assert tune.load(path) == chosen
        assert tune.get_knob("LOGRAST_DEFER_TILES") == chosen["LOGRAST_DEFER_TILES"]
    finally:
        tune.reset_knobs()

"""GPU parity tests proper: HIP kernels (through the C ABI) vs the CPU oracle on the same seeded inputs.
Integer paths must be identical; fp32 forward outputs follow the same op sequence and are compared
bit-for-bit; gradients (atomics reorder the sums) within 1e-4 relative L2 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from util import rel_l2, small_case

pytestmark = pytest.mark.gpu

GRAD_TOL = 1e-4  # relative L2, stated by BASELINE.json


def _case(name):
    from log_amd import scenes
    if name == "tiny":
        return small_case(n=150, W=48, H=40, seed=0)
    if name == "ragged":      # image not a multiple of the tile, mixed opacities
        return small_case(n=2000, W=150, H=97, focal=170.0, seed=3, smax=0.08)
    if name == "opaque":
        return small_case(n=3000, W=128, H=128, focal=150.0, seed=4, opacity=0.999, smax=0.05)
    if name == "c1":          # BASELINE config C1 geometry: 50k Gaussians, 400x400
        cams = scenes.orbit_cameras(2, W=400, H=400, focal=445.0)
        return cams[1], scenes.random_scene(50000, seed=0)
    if name == "big_splats":  # screen-filling Gaussians: wave-cooperative fill, long lists
        return small_case(n=600, W=256, H=192, focal=200.0, seed=5, smax=1.5)
    if name == "dense_tile":  # >1024 and >8192 entries in single tiles: large/huge sort classes
        cam, sc = small_case(n=12000, W=64, H=64, focal=70.0, seed=6, smax=0.01)
        sc["xyz"] *= 0.05
        return cam, sc
    if name == "huge_tiles":  # several lists beyond one LDS block (8192): hybrid multi-block sort, 1..3 merge phases
        cam, sc = small_case(n=70000, W=96, H=64, focal=100.0, seed=8, smax=0.004)
        n = [40000, 20000, 9000, 1000]
        centers = np.array([[0.0, -0.2, 0.1], [0.0, 0.25, -0.1], [0.0, 0.0, -0.25], [0.0, 0.1, 0.3]], np.float32)
        start = 0
        for c, k in zip(centers, n):
            sc["xyz"][start:start + k] = sc["xyz"][start:start + k] * 0.02 + c
            start += k
        return cam, sc
    if name == "giant_tile":  # one list of ~300 K keys (the scan's length buckets above 4096 are 4096 wide then) next to
        # many lists around the 1024-key class boundary: every list above 1024 keys must still lie inside the
        # capacity / 1024 + 1 workgroups lr_launch_sort gives the long-list kernel
        cam, sc = small_case(n=400000, W=160, H=112, focal=150.0, seed=13, smax=0.003)
        sc["xyz"][:300000] = sc["xyz"][:300000] * 0.004 + np.array([0.0, 0.12, -0.07], np.float32)
        return cam, sc
    if name == "mid_rects":   # nearly every Gaussian holds a rect of 5..16 tiles: ranked by the projection (rank rows for one
        # in four Gaussians of a batch: the rest of every batch overflows to the cursors), expanded by the wave in the fill
        cam, sc = small_case(n=30000, W=640, H=400, focal=500.0, radius=3.0, seed=21, smax=0.02)
        rng = np.random.default_rng(22)
        sc["scaling"] = (0.025 + 0.02 * rng.random((30000, 3))).astype(np.float32)
        return cam, sc
    if name.startswith("flat_depth_"):
        # all Gaussians at (almost) the same view depth, n of them inside one or two tiles: the depth-bucket sorts overflow
        # their buckets and every size class must fall back to the network (LDS class, long LDS class, hybrid)
        n = int(name.split("_")[-1])
        cam, sc = small_case(n=n, W=64, H=64, focal=70.0, seed=11, smax=0.01)
        sc["xyz"] *= 0.05
        Wm = np.asarray(cam["world_view_transform"], np.float64)
        pv = sc["xyz"].astype(np.float64) @ Wm[:3, :3] + Wm[3, :3]
        levels = np.random.default_rng(12).integers(0, 3, size=n)             # three distinct depths, many ties
        pv[:, 2] = pv[:, 2].mean() + 1e-3 * levels
        sc["xyz"] = ((pv - Wm[3, :3]) @ np.linalg.inv(Wm[:3, :3])).astype(np.float32)
        return cam, sc
    raise KeyError(name)


CASES = ["tiny", "ragged", "opaque", "c1", "big_splats", "mid_rects", "dense_tile", "huge_tiles", "giant_tile", "flat_depth_3000",
         "flat_depth_6000", "flat_depth_14000"]


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("flavour_name", ["wodilate", "upstream"])
def test_forward_bit_exact(oracle_mod, name, flavour_name):
    from log_amd import rasterizer as R
    import gpu_util as G
    flavour = R.WODILATE if flavour_name == "wodilate" else R.UPSTREAM
    cam, sc = _case(name)
    bg = (0.3, 0.6, 0.9)
    hf = G.hip_forward(cam, sc, bg, flavour)
    _, of = G.oracle_forward(oracle_mod, cam, sc, bg, flavour)
    st = G.compare_forward(hf, of)
    assert of["I"] > 0
    for k in ("radii_mismatch", "rec_bits_mismatch", "offsets_mismatch", "list_mismatch", "n_contrib_mismatch",
              "image_bits_mismatch", "final_T_bits_mismatch"):
        assert st[k] == 0, (k, st)
    if flavour.extras:
        assert st["pid_mismatch"] == 0 and st["pwp_max_abs"] == 0.0 and st["pw_max_abs"] == 0.0, st


@pytest.mark.parametrize("name", ["ragged", "c1", "big_splats", "mid_rects", "huge_tiles"])
def test_exact_reference_lists_without_tile_cull(oracle_mod, name):
    """With the binning-stage support cull off, the tile lists ARE the reference's rect lists: offsets, order and
    n_contrib identical to the oracle's.  With it on (default) they are a result-preserving subset -- checked by
    compare_forward in every other test -- and something is actually culled."""
    from log_amd import rasterizer as R
    import gpu_util as G
    cam, sc = _case(name)
    bg = (0.3, 0.6, 0.9)
    _, of = G.oracle_forward(oracle_mod, cam, sc, bg)
    prev = R.set_tile_cull(False)
    try:
        hf = G.hip_forward(cam, sc, bg)
        n_inst, _, _, n_rect = R.last_state_info()
    finally:
        R.set_tile_cull(prev)
    st = G.compare_forward(hf, of)
    assert hf["I"] == of["I"] == n_inst == n_rect
    for k in ("offsets_mismatch", "list_mismatch", "n_contrib_mismatch", "image_bits_mismatch", "pid_mismatch"):
        assert st[k] == 0, (k, st)
    hf2 = G.hip_forward(cam, sc, bg)
    n_inst, _, _, n_rect = R.last_state_info()
    assert n_rect == of["I"] and n_inst == hf2["I"] <= n_rect
    if name in ("c1", "big_splats"):
        assert hf2["I"] < of["I"]          # elongated splats: part of the rect can never reach the alpha floor
    assert (hf2["image"].view(np.uint32) == hf["image"].view(np.uint32)).all()
    # gradients are the same sums with the never-contributing entries left out
    dL = np.random.default_rng(0).standard_normal(hf["image"].shape).astype(np.float32)
    g_on = G.hip_backward(hf2, dL)
    prev = R.set_tile_cull(False)
    try:
        g_off = G.hip_backward(hf, dL)
    finally:
        R.set_tile_cull(prev)
    for k in ("conic", "means2D", "colors", "opacities"):
        assert rel_l2(g_on[k], g_off[k]) < 1e-5, k


@pytest.mark.parametrize("kind", ["closed", "open", "mixed"])
@pytest.mark.parametrize("form", ["quadrant", "rows"])
def test_lazily_ordered_lists_both_ways(oracle_mod, kind, form):
    """Lists of more than 4096 keys are ordered over their first window (7680 positions) before the compositing pass and
    to their end only where a pixel was still open there (include/lograst.h: lograst_ordered_lengths).  `closed`: screen-
    filling opaque splats -- every pixel stops after a few entries, the tails stay unordered (and the image, maps and
    gradients are the oracle's all the same).  `open`: the same lists with faint pinpoint splats -- no pixel ever stops, the
    second sort + compositing pair must order the tails and the parked waves go on where they stopped.  `mixed`: the same
    plus opaque blobs over part of the image in the middle of the depth range -- waves park with some pixels stopped and
    others open.  Both compositing forms; with the knob off the outputs are the same bits."""
    from log_amd import tune
    import gpu_util as G
    rng = np.random.default_rng(31)
    n = 11000
    cam, sc = small_case(n=n, W=64, H=48, focal=70.0, seed=30, opacity=(0.999 if kind == "closed" else 0.05),
                         smax=(0.6 if kind == "closed" else 0.004))
    if kind == "closed":
        sc["scaling"] = (0.3 + 0.3 * rng.random((n, 3))).astype(np.float32)     # every rect covers every tile
    else:
        sc["xyz"] = (sc["xyz"] * 0.06).astype(np.float32)                        # all of them inside the middle tiles
    if kind == "mixed":
        k = 60
        sc["xyz"][:k] = (0.02 * rng.standard_normal((k, 3)) + np.array([0.03, 0.0, 0.0])).astype(np.float32)
        sc["scaling"][:k] = 0.05
        sc["opacity"][:k] = 0.999
    bg = (0.3, 0.6, 0.9)
    _, of = G.oracle_forward(oracle_mod, cam, sc, bg)
    hf = G.hip_forward(cam, sc, bg, fwd_form=form, scratch_floats=16)
    lens = np.diff(hf["tile_offsets"].astype(np.int64))
    long_lists = int((lens > 7680).sum())
    assert long_lists > 0, lens.max()
    if kind == "closed":
        assert hf["lazy_lists"] == long_lists and (hf["ordered_len"][lens > 7680] <= 7680).all()
    else:
        assert hf["lazy_lists"] < long_lists                                     # somebody asked for the tails
        assert (hf["n_contrib"].astype(np.int64).max() > 7680)                   # ... because the walk went there
        if kind == "mixed":   # inside a tile whose walk went into the tail, some pixels had stopped before it (alpha <= 0.99: a stopped pixel has T < 0.01)
            H, W = hf["n_contrib"].shape
            gx = (W + 15) // 16
            ys, xs = np.mgrid[0:H, 0:W]
            tile = (ys // 16) * gx + xs // 16
            deep = np.zeros(len(lens), bool)
            deep[np.unique(tile[hf["n_contrib"] > 7680])] = True
            stopped_early = deep[tile] & (hf["final_T"] < 0.01) & (hf["n_contrib"] < 7000)   # (a first window ends past 7648)
            assert stopped_early.sum() > 10, int(stopped_early.sum())
    st = G.compare_forward(hf, of)
    for k in ("radii_mismatch", "rec_bits_mismatch", "offsets_mismatch", "list_mismatch", "n_contrib_mismatch",
              "image_bits_mismatch", "final_T_bits_mismatch", "pid_mismatch"):
        assert st[k] == 0, (k, st)
    assert st["pwp_max_abs"] == 0.0 and st["pw_max_abs"] == 0.0, st
    dL = np.random.default_rng(2).standard_normal(hf["image"].shape).astype(np.float32)
    g = G.hip_backward(hf, dL, bwd_form=form)        # (same form: its visits come from the forward's hit masks, written by
    assert g["bwd_masks"]                             # both compositing passes where waves parked)
    og = oracle_mod.backward(of["_view"], of, dL)
    for k in ("means2D", "conic", "opacities", "colors"):
        assert rel_l2(g[k], og[k]) < GRAD_TOL, (k, rel_l2(g[k], og[k]))
    tune.set_knob("LOGRAST_LAZY_SORT", 0)
    try:
        hf0 = G.hip_forward(cam, sc, bg, fwd_form=form, scratch_floats=16)
        assert hf0["lazy_lists"] == 0
        for k in ("image", "final_T", "point_weight_pixel", "point_weight"):
            assert (hf0[k].view(np.uint32) == hf[k].view(np.uint32)).all(), k
        for k in ("radii", "tile_offsets", "point_list", "n_contrib", "point_id_pixel"):
            assert (hf0[k] == hf[k]).all(), k
        g0 = G.hip_backward(hf0, dL, bwd_form=form)
    finally:
        tune.reset_knobs()
    for k in ("means2D", "conic", "opacities", "colors"):
        assert rel_l2(g[k], g0[k]) < 1e-5, k


def test_backward_reads_only_the_ordered_part_of_a_lazily_ordered_list(oracle_mod):
    """The reverse walk of a view whose lists were left at their first window (the product path: nobody calls
    lograst_finish_lists there) against the oracle's gradients: through the autograd Function, whose forward keeps no keys
    and finishes nothing."""
    from diff_gaussian_rasterization_wodilate import GaussianRasterizer
    import gpu_util as G
    rng = np.random.default_rng(31)
    n = 11000
    cam, sc = small_case(n=n, W=64, H=48, focal=70.0, seed=30, opacity=0.999, smax=0.6)
    sc["scaling"] = (0.3 + 0.3 * rng.random((n, 3))).astype(np.float32)
    bg = (0.3, 0.6, 0.9)
    dev = torch.device("cuda:0")
    v, of = G.oracle_forward(oracle_mod, cam, sc, bg)
    assert np.diff(of["tile_offsets"].astype(np.int64)).max() > 7680
    dL = np.random.default_rng(2).standard_normal(of["image"].shape).astype(np.float32)
    og = oracle_mod.backward(v, of, dL)
    t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev).requires_grad_(True)
    leaves = dict(means3D=t(sc["xyz"]), scales=t(sc["scaling"]), rotations=t(sc["rotation"]),
                  opacities=t(sc["opacity"]), colors=t(sc["colors"]))
    rast = GaussianRasterizer(raster_settings=G.settings(cam, bg, dev))
    means2D = torch.zeros(n, 3, device=dev, requires_grad=True)
    out = rast(means3D=leaves["means3D"], means2D=means2D, shs=None, colors_precomp=leaves["colors"],
               opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
    assert (out[0].detach().cpu().numpy().view(np.uint32) == of["image"].view(np.uint32)).all()
    out[0].backward(gradient=torch.from_numpy(dL).to(dev))
    assert rel_l2(leaves["colors"].grad.cpu().numpy(), og["colors"]) < GRAD_TOL
    assert rel_l2(leaves["opacities"].grad.cpu().numpy().reshape(-1), np.asarray(og["opacities"]).reshape(-1)) < GRAD_TOL
    assert rel_l2(means2D.grad.cpu().numpy()[:, :2], np.asarray(og["means2D"])[:, :2]) < GRAD_TOL


def _flavour(name):
    from log_amd import rasterizer as R
    return {"wodilate": R.WODILATE, "upstream": R.UPSTREAM}[name]


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("flavour_name", ["wodilate", "upstream"])
@pytest.mark.parametrize("path", ["plain", "training"])
def test_backward_vs_oracle(oracle_mod, name, flavour_name, path):
    """Both packages' backward (round-5 verdict, next #2a: the upstream flavour -- `use_origin_render: True`,
    LoG/render/renderer.py:99-107, cov + 0.3 with a transparent gradient, LoG/model/geometry.py:87-88 -- had no backward case
    through scales / rotations), each as a plain backend call and as the TRAINING forward the autograd path makes
    (accumulator rows prepared by the forward, the forward's hit masks handed to the reverse walk)."""
    import gpu_util as G
    cam, sc = _case(name)
    bg = (0.3, 0.6, 0.9)
    fl = _flavour(flavour_name)
    hf = G.hip_forward(cam, sc, bg, flavour=fl, scratch_floats=16 if path == "training" else 0)
    v, of = G.oracle_forward(oracle_mod, cam, sc, bg, flavour=fl)
    dL = np.random.default_rng(1).random(of["image"].shape, dtype=np.float32)
    hg = G.hip_backward(hf, dL, bwd_form=hf["fwd_form"] if path == "training" else None)
    assert hg["bwd_masks"] == (path == "training")
    og = oracle_mod.backward(v, of, dL)
    # A6, every row: outputs of the reverse walk
    for k in ("means2D", "conic", "opacities", "colors"):
        assert rel_l2(hg[k], og[k]) < GRAD_TOL, (k, rel_l2(hg[k], og[k]))
    assert (hg["means2D"][:, 2] == 0).all()
    # A6 + A6b end to end: every row against the float64 twin of the backward (tests/gpu_util.py: rel-L2 <= 1e-4 over all
    # rows the chain rule conditions to better than 500x, and on EVERY row HIP no further from float64 than twice the
    # fp32 oracle; the small scenes hold more pancake-flat Gaussians than the bench scenes: up to 10 % above the bound)
    g64 = oracle_mod.backward_f64(v, of, dL)
    G.assert_gradients_anchored(G.gradient_anchor_stats(hg, og, g64), tol=GRAD_TOL, max_excluded=0.10,
                                name="case_%s_%s_%s" % (name, flavour_name, path),
                                # the +0.3 low-pass keeps every 2-D covariance >= 0.3 I: no near-singular conics, so the plain
                                # all-rows criterion of north_star holds on these scenes too (measured: see profiles/)
                                all_rows_tol=None)


@pytest.mark.parametrize("form", ["quadrant", "rows"])
@pytest.mark.parametrize("name", ["ragged", "c1", "big_splats", "dense_tile", "huge_tiles"])
def test_hit_masks_hand_over(oracle_mod, name, form):
    """Round 6: a training forward leaves the ballots of its per-chunk support tests in lograst_view.hit_masks and the
    reverse walk of the SAME form takes its visits from them (no support test, only the visited records gathered).  Checked:
    the masked walk's gradients are the oracle's; they are what a forward without the buffer gives (same addends, atomics
    reorder the sums); a reverse walk of the OTHER form ignores the buffer (and is the oracle's too); and the masks really
    are what drives the visits -- with the buffer zeroed the masked walk visits nothing."""
    import gpu_util as G
    cam, sc = _case(name)
    bg = (0.3, 0.6, 0.9)
    other = "rows" if form == "quadrant" else "quadrant"
    v, of = G.oracle_forward(oracle_mod, cam, sc, bg)
    dL = np.random.default_rng(4).standard_normal(of["image"].shape).astype(np.float32)
    og = oracle_mod.backward(v, of, dL)
    hf = G.hip_forward(cam, sc, bg, scratch_floats=16, fwd_form=form)
    saved = hf["_torch"][-1]
    assert saved["hit_masks"] is not None and saved["hit_mask_form"] == {"rows": 1, "quadrant": 2}[form]
    g = G.hip_backward(hf, dL, bwd_form=form)
    assert g["bwd_masks"]
    g_other = G.hip_backward(hf, dL, bwd_form=other)
    assert not g_other["bwd_masks"]
    hf0 = G.hip_forward(cam, sc, bg, scratch_floats=16, fwd_form=form, hit_masks=False)
    assert hf0["_torch"][-1]["hit_masks"] is None
    for k in ("image", "final_T", "point_weight"):
        assert (hf0[k].view(np.uint32) == hf[k].view(np.uint32)).all(), k
    assert (hf0["n_contrib"] == hf["n_contrib"]).all() and (hf0["point_list"] == hf["point_list"]).all()
    g0 = G.hip_backward(hf0, dL, bwd_form=form)
    assert not g0["bwd_masks"]
    for k in ("means2D", "conic", "opacities", "colors"):
        assert rel_l2(g[k], og[k]) < GRAD_TOL, (k, rel_l2(g[k], og[k]))
        assert rel_l2(g_other[k], og[k]) < GRAD_TOL, (k, rel_l2(g_other[k], og[k]))
        assert rel_l2(g[k], g0[k]) < 1e-5, (k, rel_l2(g[k], g0[k]))
    for k in ("means3D", "scales", "rotations"):
        assert rel_l2(g[k], g0[k]) < 1e-4, (k, rel_l2(g[k], g0[k]))
    saved["hit_masks"].zero_()
    gz = G.hip_backward(hf, dL, bwd_form=form)
    assert gz["bwd_masks"] and not gz["colors"].any() and not gz["conic"].any() and not gz["opacities"].any()
    assert G.hip_backward(hf, dL, bwd_form=other)["colors"].any()      # the other form never looked at the buffer


@pytest.mark.parametrize("flavour_name", ["wodilate", "upstream"])
def test_backward_with_scale_modifier(oracle_mod, flavour_name):
    """`scale_modifier` != 1 through the backward (round-5 verdict, next #2c: the forward-only case below was the only one):
    the chain rule to dL/dscales carries the factor, dL/drotations sees the scaled covariance."""
    import gpu_util as G
    cam, sc = _case("ragged")
    bg = (0.3, 0.6, 0.9)
    fl = _flavour(flavour_name)
    hf = G.hip_forward(cam, sc, bg, flavour=fl, scale_modifier=1.7, scratch_floats=16)
    v, of = G.oracle_forward(oracle_mod, cam, sc, bg, flavour=fl, scale_modifier=1.7)
    st = G.compare_forward(hf, of)
    for k in ("radii_mismatch", "rec_bits_mismatch", "image_bits_mismatch", "n_contrib_mismatch"):
        assert st[k] == 0, (k, st)
    dL = np.random.default_rng(7).random(of["image"].shape, dtype=np.float32)
    hg = G.hip_backward(hf, dL)
    og = oracle_mod.backward(v, of, dL)
    for k in ("means2D", "conic", "opacities", "colors"):
        assert rel_l2(hg[k], og[k]) < GRAD_TOL, (k, rel_l2(hg[k], og[k]))
    hp = G.hip_project_backward(hf, og["means2D"], og["conic"])
    for k in ("means3D", "scales", "rotations"):
        assert rel_l2(hp[k], og[k]) < 1e-6, (k, rel_l2(hp[k], og[k]))
    g64 = oracle_mod.backward_f64(v, of, dL)
    G.assert_gradients_anchored(G.gradient_anchor_stats(hg, og, g64), tol=GRAD_TOL, max_excluded=0.10,
                                name="scale_modifier_" + flavour_name)
    # and the factor is really in there: against the same scene at scale_modifier = 1 the scale gradients differ
    hf1 = G.hip_forward(cam, sc, bg, flavour=fl, scale_modifier=1.0, scratch_floats=16)
    assert rel_l2(G.hip_backward(hf1, dL)["scales"], hg["scales"]) > 1e-2


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("flavour_name", ["wodilate", "upstream"])
def test_project_backward_isolated(oracle_mod, name, flavour_name):
    """A6b alone on identical inputs (the oracle's dL/dmean2D, dL/dconic): same op sequence, so the
    kernel must agree with the oracle to fp32 rounding on EVERY row, ill-conditioned ones included.  Both low-pass
    flavours: the fork's clamp (masked sub-gradient) and the upstream +0.3 (transparent)."""
    import gpu_util as G
    cam, sc = _case(name)
    bg = (0.3, 0.6, 0.9)
    fl = _flavour(flavour_name)
    hf = G.hip_forward(cam, sc, bg, flavour=fl)
    v, of = G.oracle_forward(oracle_mod, cam, sc, bg, flavour=fl)
    dL = np.random.default_rng(1).random(of["image"].shape, dtype=np.float32)
    og = oracle_mod.backward(v, of, dL)
    hg = G.hip_project_backward(hf, og["means2D"], og["conic"])
    for k in ("means3D", "scales", "rotations"):
        assert rel_l2(hg[k], og[k]) < 1e-6, (k, rel_l2(hg[k], og[k]))


def test_use_filter_false_and_scale_modifier(oracle_mod):
    from log_amd import rasterizer as R
    import gpu_util as G
    cam, sc = _case("ragged")
    for kw in (dict(use_filter=False), dict(scale_modifier=1.7)):
        hf = G.hip_forward(cam, sc, (0, 0, 0), R.WODILATE, **kw)
        _, of = G.oracle_forward(oracle_mod, cam, sc, (0, 0, 0), R.WODILATE, **kw)
        st = G.compare_forward(hf, of)
        assert st["radii_mismatch"] == 0 and st["list_mismatch"] == 0 and st["image_bits_mismatch"] == 0, (kw, st)


def test_compute_radius_bit_exact(oracle_mod):
    """A0 through the LoG.cuda drop-in module vs the oracle, on the golden inputs from the reference."""
    import glob, math, os
    from log_amd.compute_radius import compute_radius_module
    dev = torch.device("cuda:0")
    for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "geom_*.npz"))):
        g = np.load(path)
        W, H = int(g["W"]), int(g["H"])
        tfx, tfy = math.tan(float(g["FoVx"]) * 0.5), math.tan(float(g["FoVy"]) * 0.5)
        fx, fy = W / (2 * tfx), H / (2 * tfy)
        t = lambda a: torch.tensor(a, device=dev)
        r = compute_radius_module.compute_radius(t(g["xyz"]), t(g["scaling"]), t(g["rotation"]),
                                                 t(g["full_proj_transform"]), t(g["world_view_transform"]),
                                                 fx, fy, tfx, tfy).cpu().numpy()
        o = oracle_mod.compute_radius(g["xyz"], g["scaling"], g["rotation"], g["full_proj_transform"],
                                      g["world_view_transform"], fx, fy, tfx, tfy)
        assert (r.view(np.uint32) == o.view(np.uint32)).all(), np.abs(r - o).max()
        kept = r > 0
        np.testing.assert_allclose(r[kept], g["ref_radius_clamp"][kept], rtol=2e-4, atol=1e-4)


def test_empty_and_all_culled(oracle_mod):
    import gpu_util as G
    cam, sc = _case("tiny")
    empty = {k: v[:0] for k, v in sc.items()}
    hf = G.hip_forward(cam, empty, (0.1, 0.2, 0.3))
    assert hf["I"] == 0 and (hf["n_contrib"] == 0).all() and (hf["point_id_pixel"] == -1).all()
    np.testing.assert_array_equal(hf["image"][1], np.float32(0.2))
    behind = dict(sc)
    behind["xyz"] = (sc["xyz"] + 3.0 * cam["camera_center"][None]).astype(np.float32)   # beyond the camera, behind it
    hf = G.hip_forward(cam, behind, (0.1, 0.2, 0.3))
    _, of = G.oracle_forward(oracle_mod, cam, behind, (0.1, 0.2, 0.3))
    assert hf["I"] == of["I"] and (hf["radii"] == of["radii"]).all()
    hg = G.hip_backward(hf, np.ones_like(hf["image"]))
    assert all(np.abs(v).max() == 0 for v in hg.values())


def test_backward_scratch_lifecycle():
    """The forward zero-fills the backward's accumulators when a gradient is wanted; a second backward through the
    same graph (retain_graph) must not reuse the consumed block, and a no_grad forward allocates none."""
    from diff_gaussian_rasterization_wodilate import GaussianRasterizer
    import gpu_util as G
    cam, sc = _case("ragged")
    dev = torch.device("cuda:0")
    rs = G.settings(cam, (0.2, 0.3, 0.4), dev)
    t = lambda a, g=True: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev, requires_grad=g)
    leaves = dict(means3D=t(sc["xyz"]), colors_precomp=t(sc["colors"]), opacities=t(sc["opacity"]),
                  scales=t(sc["scaling"]), rotations=t(sc["rotation"]))
    m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
    rast = GaussianRasterizer(raster_settings=rs)
    w = torch.rand(3, cam["image_height"], cam["image_width"], device=dev)
    img = rast(means2D=m2, shs=None, cov3D_precomp=None, **leaves)[0]
    img.backward(gradient=w, retain_graph=True)
    first = {k: v.grad.clone() for k, v in leaves.items()}
    first_m2 = m2.grad.clone()
    for v in leaves.values():
        v.grad = None
    m2.grad = None
    img.backward(gradient=w)
    for k, v in leaves.items():
        assert rel_l2(v.grad.cpu().numpy(), first[k].cpu().numpy()) < 1e-5, k
    assert rel_l2(m2.grad.cpu().numpy(), first_m2.cpu().numpy()) < 1e-5
    with torch.no_grad():
        img2 = rast(means2D=m2, shs=None, cov3D_precomp=None, **leaves)[0]
    assert (img2 == img).all() and not img2.requires_grad


def test_module_autograd_contract():
    """The nn.Module / autograd surface LoG drives (renderer.py:135-165,190-198; counter.py:40,46)."""
    from diff_gaussian_rasterization_wodilate import GaussianRasterizer
    import diff_gaussian_rasterization as up
    import gpu_util as G
    cam, sc = _case("ragged")
    dev = torch.device("cuda:0")
    rs = G.settings(cam, (1, 1, 1), dev)
    T = lambda a: torch.tensor(a, device=dev, requires_grad=True)
    m3, sca, rot, op, col = (T(sc[k]) for k in ("xyz", "scaling", "rotation", "opacity", "colors"))
    m2 = torch.zeros_like(m3, requires_grad=True)
    m2.retain_grad()
    rast = GaussianRasterizer(raster_settings=rs)
    assert rast.raster_settings.image_width == cam["image_width"]
    kw = dict(means3D=m3, means2D=m2, shs=None, colors_precomp=col, opacities=op, scales=sca, rotations=rot,
              cov3D_precomp=None)
    ret = rast(**kw)
    assert len(ret) == 5
    image, radii, pid, pwp, pw = ret
    assert image.shape == (3, cam["image_height"], cam["image_width"]) and radii.dtype == torch.int32
    assert pid.shape == image.shape[1:] and pw.shape == (len(sc["xyz"]),) and not pw.requires_grad
    ids, counts = torch.unique(pid, sorted=True, return_counts=True)      # renderer.py:156
    assert ids[0] == -1 or ids[0] >= 0
    image.sum().backward()
    g1 = {k: t.grad.clone() for k, t in dict(m3=m3, m2=m2, sca=sca, rot=rot, op=op, col=col).items()}
    assert g1["op"].shape == op.shape and g1["m2"].shape == m2.shape
    assert float(g1["m2"][:, :2].abs().sum()) > 0 and float(g1["m2"][:, 2].abs().sum()) == 0
    # second pass through the same rasterizer / same leaves accumulates (depth pass, renderer.py:186-201)
    rast(**kw)[0].sum().backward()
    for k, t in dict(m3=m3, m2=m2, sca=sca, rot=rot, op=op, col=col).items():
        # the two passes are separate atomic sums: compare in relative L2 (elementwise, ill-conditioned rows flake)
        assert rel_l2(t.grad.cpu().numpy(), (2 * g1[k]).cpu().numpy()) < 1e-4, k
    # eval-mode kwarg of the fork and the 2-tuple upstream flavour
    with torch.no_grad():
        assert len(rast(use_filter=False, **kw)) == 5
        ret2 = up.GaussianRasterizer(raster_settings=rs)(**kw)
        assert len(ret2) == 2
        with pytest.raises(TypeError):
            up.GaussianRasterizer(raster_settings=rs)(use_filter=False, **kw)
        rad = rast.compute_radius(m3, sca, rot)
        assert rad.shape == (len(sc["xyz"]),) and rad.dtype == torch.float32
    with pytest.raises(Exception, match="excatly one"):
        rast(means3D=m3, means2D=m2, shs=None, colors_precomp=None, opacities=op, scales=sca, rotations=rot)


def test_deterministic_forward():
    import gpu_util as G
    cam, sc = _case("opaque")
    a = G.hip_forward(cam, sc, (0, 0, 0))
    b = G.hip_forward(cam, sc, (0, 0, 0))
    assert (a["image"].view(np.uint32) == b["image"].view(np.uint32)).all()
    assert (a["point_list"] == b["point_list"]).all()


def test_capacity_hint_and_overflow_flag():
    from log_amd import rasterizer as R
    import gpu_util as G
    cam, sc = _case("ragged")
    exact = G.hip_forward(cam, sc, (0, 0, 0))
    try:
        R.set_instance_capacity(exact["I"] + 1000)
        hinted = G.hip_forward(cam, sc, (0, 0, 0))
        n, over = R.last_overflow()
        assert n == exact["I"] and not over
        assert (hinted["image"].view(np.uint32) == exact["image"].view(np.uint32)).all()
        R.set_instance_capacity(max(exact["I"] // 2, 1))
        G.hip_forward(cam, sc, (0, 0, 0))
        n, over = R.last_overflow()
        assert n == exact["I"] and over
    finally:
        R.set_instance_capacity(None)


def test_fused_multi_view_accumulation_matches_autograd():
    """log_amd.rasterizer.accumulate_grads_into (LOGRAST_BWD_ACCUMULATE): two views added straight into a
    GradientBucket == the sum autograd accumulates view by view."""
    from diff_gaussian_rasterization_wodilate import GaussianRasterizer
    from log_amd import rasterizer as R, scenes
    from log_amd.dist import GradientBucket
    import gpu_util as G
    dev = torch.device("cuda:0")
    cams = scenes.orbit_cameras(3, W=160, H=112, focal=170.0)
    sc = scenes.random_scene(4000, seed=11, opacity=None, smax=0.07)
    n = 4000
    w = torch.tensor(np.random.default_rng(3).random((3, 112, 160), dtype=np.float32), device=dev)
    names = dict(means3D="xyz", scales="scaling", rotations="rotation", opacities="opacity", colors="colors")

    def run(fused):
        leaves = {k: torch.tensor(sc[v], device=dev, requires_grad=True) for k, v in names.items()}
        bucket = GradientBucket(n, dev)
        bucket.attach(leaves)
        m2s = []
        for cam in cams[:2]:
            rast = GaussianRasterizer(raster_settings=G.settings(cam, (1, 1, 1), dev))
            m2 = torch.zeros(n, 3, device=dev, requires_grad=True)
            kw = dict(means3D=leaves["means3D"], means2D=m2, shs=None, colors_precomp=leaves["colors"],
                      opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
                      cov3D_precomp=None)
            if fused:
                with R.accumulate_grads_into(bucket.views):
                    (rast(**kw)[0] * w).sum().backward()
            else:
                (rast(**kw)[0] * w).sum().backward()
            m2s.append(m2.grad.clone())
        torch.cuda.synchronize()
        return bucket.flat.clone(), m2s

    prev = R.set_inplace_leaf_grads(False)     # the comparison side: autograd's own accumulation, view by view
    try:
        a, m2a = run(False)
    finally:
        R.set_inplace_leaf_grads(prev)
    b, m2b = run(True)
    assert float(a.abs().sum()) > 0
    assert rel_l2(b.cpu().numpy(), a.cpu().numpy()) < 1e-5
    for x, y in zip(m2a, m2b):
        assert rel_l2(y.cpu().numpy(), x.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("W,H", [(2560, 1440), (3840, 2160), (7680, 4320)],
                         ids=["1440p_batched", "4k_scan32_batched_130KB_lds", "8k_scan128_unbatched"])
def test_large_tile_grids(oracle_mod, W, H):
    """Tile counts beyond 8192 use the wider scan variants (32 / 128 tiles per thread); up to 40000 tiles (4K) the
    batched projection still ranks in LDS (more than 64 KB of dynamic LDS at 4K), beyond that the unbatched kernel runs."""
    import gpu_util as G
    from log_amd import scenes
    cam = scenes.orbit_cameras(1, W=W, H=H, focal=2139.0 * W / 1920.0)[0]
    sc = scenes.random_scene(20000, seed=9, opacity=None, smax=0.02)
    hf = G.hip_forward(cam, sc, (0.5, 0.5, 0.5))
    _, of = G.oracle_forward(oracle_mod, cam, sc, (0.5, 0.5, 0.5))
    st = G.compare_forward(hf, of)
    for k in ("radii_mismatch", "offsets_mismatch", "list_mismatch", "n_contrib_mismatch", "image_bits_mismatch",
              "pid_mismatch"):
        assert st[k] == 0, (k, st)


@pytest.mark.parametrize("package", ["wodilate", "upstream"])
def test_cov3d_precomp_vs_oracle(oracle_mod, package):
    """The module's `cov3D_precomp` input (third-party forward signature; LoG itself passes scales + rotations,
    renderer.py:134,149): the kernels read the [N, 6] covariances instead of computing them -- image / radii / fork maps
    bit-identical to the oracle's cov3D path, dL/dcov3D and the other gradients within 1e-4, scales / rotations absent
    from the graph, scale_modifier without effect."""
    import diff_gaussian_rasterization as up
    import diff_gaussian_rasterization_wodilate as wo
    import gpu_util as G
    from log_amd import rasterizer as R
    from oracle import torch_oracle
    cam, sc = _case("ragged")
    dev = torch.device("cuda:0")
    Rm = torch_oracle._rot(torch.tensor(sc["rotation"], dtype=torch.float64))
    M = Rm * torch.tensor(sc["scaling"], dtype=torch.float64)[:, None, :]
    S = (M @ M.transpose(1, 2)).numpy()
    cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], axis=1).astype(np.float32)
    mod, flavour = (wo, R.WODILATE) if package == "wodilate" else (up, R.UPSTREAM)
    bg = (0.2, 0.5, 0.1)
    from util import cam_tan
    tfx, tfy = cam_tan(cam)
    v = oracle_mod.make_view(cam["image_width"], cam["image_height"], tfx, tfy, cam["world_view_transform"],
                             cam["full_proj_transform"], bg, filter_mode=flavour.filter_mode, ndc_cull=flavour.ndc_cull)
    of = oracle_mod.forward(v, sc["xyz"], None, None, sc["opacity"], sc["colors"], cov3d=cov, extras=bool(flavour.extras))
    dL = np.random.default_rng(4).random(of["image"].shape, dtype=np.float32)
    og = oracle_mod.backward(v, of, dL)
    T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev, requires_grad=True)
    m3, op, col, cv = T(sc["xyz"]), T(sc["opacity"]), T(sc["colors"]), T(cov)
    m2 = torch.zeros_like(m3, requires_grad=True)
    rast = mod.GaussianRasterizer(raster_settings=G.settings(cam, bg, dev, scale_modifier=0.37))   # (no effect here)
    out = rast(means3D=m3, means2D=m2, shs=None, colors_precomp=col, opacities=op, scales=None, rotations=None,
               cov3D_precomp=cv)
    assert len(out) == (5 if flavour.extras else 2)
    assert np.array_equal(out[0].detach().cpu().numpy(), of["image"])
    assert np.array_equal(out[1].cpu().numpy(), of["radii"])
    if flavour.extras:
        assert np.array_equal(out[2].cpu().numpy(), of["point_id_pixel"])
        assert np.array_equal(out[4].cpu().numpy(), of["point_weight"])
    out[0].backward(gradient=torch.tensor(dL, device=dev))
    assert cv.grad.shape == (len(cov), 6)
    for name, leaf in (("cov3D", cv), ("means3D", m3), ("opacities", op), ("colors", col), ("means2D", m2)):
        assert rel_l2(leaf.grad.cpu().numpy().reshape(og[name].shape), og[name]) < 1e-4, name
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=m3, means2D=m2, shs=None, colors_precomp=col, opacities=op, scales=T(sc["scaling"]),
             rotations=None, cov3D_precomp=cv)
    with pytest.raises(ValueError, match=r"\[N, 6\]"):
        rast(means3D=m3, means2D=m2, shs=None, colors_precomp=col, opacities=op, scales=None, rotations=None,
             cov3D_precomp=cv[:, :5])


def test_psnr_against_the_float64_twin(oracle_mod):
    """"image PSNR vs oracle / reference <= 0.05 dB apart".  Against the C oracle the image is bit-identical
    (test_forward_bit_exact, incl. the C1 geometry: PSNR = inf); against the independent float64 autograd restatement
    (oracle/torch_oracle.py) the fp32 kernels' image differs by rounding only: PSNR far above any visible level."""
    import gpu_util as G
    from oracle import torch_oracle
    from util import cam_tan
    cam, sc = _case("ragged")
    bg = (0.3, 0.6, 0.9)
    hf = G.hip_forward(cam, sc, bg)
    T = lambda a: torch.tensor(a, dtype=torch.float64)
    tfx, tfy = cam_tan(cam)
    img, radii, _ = torch_oracle.render(
        cam["image_width"], cam["image_height"], tfx, tfy, T(cam["world_view_transform"]), T(cam["full_proj_transform"]),
        T(bg), T(sc["xyz"]), torch.zeros(len(sc["xyz"]), 3, dtype=torch.float64), T(sc["scaling"]), T(sc["rotation"]),
        T(sc["opacity"]), T(sc["colors"]), filter_mode=2, ndc_cull=True)
    mse = float(((torch.tensor(hf["image"], dtype=torch.float64) - img) ** 2).mean())
    psnr = 10.0 * np.log10(1.0 / max(mse, 1e-30))
    assert psnr > 100.0, psnr
    assert (radii.numpy() == hf["radii"]).all()

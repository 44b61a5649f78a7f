"""BASELINE.json full-size configurations on the MI355X: oracle comparison where the oracle finishes in
seconds (C2: 1 M Gaussians @1080p), size-independent properties beyond that (4K image, 10 M Gaussians):
sortedness of every tile list, instance conservation, determinism, linearity / homogeneity of the backward."""
import math

import numpy as np
import pytest
import torch

from util import cam_tan, rel_l2

pytestmark = pytest.mark.gpu


def _scene(n, W, H, seed=0, opacity=0.999, view=0, n_views=8):
    from log_amd import scenes
    cams = scenes.orbit_cameras(n_views, W=W, H=H, focal=2139.0 * W / 1920.0)
    return cams[view], scenes.random_scene(n, seed=seed, opacity=opacity)


def _tile_sorted_ok(saved, W, H, dev):
    """Every tile's slice of point_list is strictly ascending in (depth bits, id)."""
    from log_amd import rasterizer as R
    offs = R.tile_offsets_of(saved, W, H).to(torch.int64)
    I = int(offs[-1])
    if I == 0:
        return True, 0
    plist = saved["plist"][:I].to(torch.int64)
    depth = saved["geom"][:16 * saved["radii"].numel()].view(-1, 16)[:, 9].view(torch.int32).to(torch.int64)   # positive floats: bits are monotone
    key = depth[plist] * (1 << 32) + plist
    tile_of = torch.repeat_interleave(torch.arange(len(offs) - 1, device=dev), offs[1:] - offs[:-1])
    same = tile_of[1:] == tile_of[:-1]
    ok = bool(((key[1:] > key[:-1]) | ~same).all())
    return ok, I


def _rects_total(saved):
    g = saved["geom"][:16 * saved["radii"].numel()].view(-1, 16)
    r0 = g[:, 10].view(torch.int32)
    r1 = g[:, 11].view(torch.int32)
    w = (r1 & 0xffff) - (r0 & 0xffff)
    h = (r1 >> 16) - (r0 >> 16)
    return int((w * h).sum().item())


def test_c2_full_size_vs_oracle(oracle_mod):
    """C2 (bench workload): 1 M Gaussians, 1920x1080, view 0 -- full bit-exact forward + backward parity."""
    import gpu_util as G
    cam, sc = _scene(1_000_000, 1920, 1080)
    bg = (1.0, 1.0, 1.0)
    hf = G.hip_forward(cam, sc, bg)
    v, of = G.oracle_forward(oracle_mod, cam, sc, bg)
    st = G.compare_forward(hf, of)
    assert of["I"] > 3_000_000
    for k in ("radii_mismatch", "rec_bits_mismatch", "offsets_mismatch", "list_mismatch", "n_contrib_mismatch",
              "image_bits_mismatch", "final_T_bits_mismatch", "pid_mismatch"):
        assert st[k] == 0, (k, st)
    assert st["pwp_max_abs"] == 0.0 and st["pw_max_abs"] == 0.0
    dL = np.random.default_rng(1).random(of["image"].shape, dtype=np.float32)
    hg = G.hip_backward(hf, dL)
    og = oracle_mod.backward(v, of, dL)
    for k in ("means2D", "conic", "opacities", "colors", "means3D"):
        assert rel_l2(hg[k], og[k]) < 1e-4, (k, rel_l2(hg[k], og[k]))
    hp = G.hip_project_backward(hf, og["means2D"], og["conic"])
    for k in ("means3D", "scales", "rotations"):
        assert rel_l2(hp[k], og[k]) < 1e-6, k


def _assert_forward(hf, of, check_lists, form):
    """(the upstream package's flavour has no fork maps: those keys are simply absent on both sides)"""
    import gpu_util as G
    st = G.compare_forward(hf, of) if check_lists else None
    if st is not None:
        for k in ("radii_mismatch", "rec_bits_mismatch", "offsets_mismatch", "list_mismatch", "n_contrib_mismatch",
                  "image_bits_mismatch", "final_T_bits_mismatch", "pid_mismatch"):
            assert st.get(k, 0) == 0, (form, k, st)
        assert st.get("pwp_max_abs", 0.0) == 0.0 and st.get("pw_max_abs", 0.0) == 0.0, form
    else:
        for k in ("radii", "point_id_pixel"):
            assert k not in hf or (hf[k] == of[k]).all(), (form, k)
        for k in ("image", "final_T", "point_weight_pixel", "point_weight"):
            assert k not in hf or (hf[k].view(np.uint32) == of[k].view(np.uint32)).all(), (form, k)


def _full_parity(oracle_mod, cam, sc, bg, dL_seed=1, check_lists=True, stats_name=None,
                 fwd_forms=("rows", "quadrant"), all_rows_tol=None, flavour=None):
    """Forward bit-exact (lists: exact, or the oracle's lists minus provably invisible entries), backward <= 1e-4 rel-L2
    on every output of the reverse walk, chain rule <= 1e-6 on identical inputs.  The forward prepares the backward's
    accumulators as the autograd path does (touched-only dL/dconic on large inputs).
    fwd_forms: BOTH forms of the compositing kernel are compared with the ORACLE, bit for bit (round-3 verdict: the
    row-split form -- what bench.py's headline runs -- was only ever compared with the quadrant form); the last one's
    forward feeds the backward."""
    import gpu_util as G
    from log_amd import rasterizer as R
    flavour = flavour or R.WODILATE
    v, of = G.oracle_forward(oracle_mod, cam, sc, bg, flavour=flavour)
    dL = np.random.default_rng(dL_seed).random(of["image"].shape, dtype=np.float32)
    og = oracle_mod.backward(v, of, dL)
    hf, hg = None, None
    for form in fwd_forms:
        del hf
        hf = G.hip_forward(cam, sc, bg, flavour=flavour, scratch_floats=16, fwd_form=form)
        assert hf["fwd_form"] == form
        _assert_forward(hf, of, check_lists, form)
        # the reverse walk of the SAME form: its visits come from this forward's hit masks (round 6) -- every row of its four
        # outputs against the oracle, for both forms
        hg = G.hip_backward(hf, dL, bwd_form=form)
        assert hg["bwd_masks"], form
        for k in ("means2D", "conic", "opacities", "colors"):
            assert rel_l2(hg[k], og[k]) < 1e-4, (form, k, rel_l2(hg[k], og[k]))
    hn = G.hip_backward(hf, dL)                                     # ... and in the form the package picks by itself
    for k in ("means2D", "conic", "opacities", "colors"):          # the reverse walk: every row
        assert rel_l2(hn[k], og[k]) < 1e-4, (k, hn["bwd_form"], rel_l2(hn[k], og[k]))
    hp = G.hip_project_backward(hf, og["means2D"], og["conic"])    # the chain rule on identical inputs: every row
    for k in ("means3D", "scales", "rotations"):
        assert rel_l2(hp[k], og[k]) < 1e-6, k
    # end to end, every row, anchored on the float64 twin of the backward (tests/gpu_util.py: the four tested claims --
    # 1e-4 rel-L2 over all rows the chain rule conditions to better than 500x, HIP no further from float64 than twice the
    # fp32 oracle on EVERY row, exact zeros where the gradient is zero); no masked or quantile criterion
    g64 = oracle_mod.backward_f64(v, of, dL)
    # all_rows_tol: north_star's plain criterion on top -- rel-L2 <= tol over ALL rows of dL/dmeans3D / dL/dscales /
    # dL/drotations end to end, HIP vs float64 and HIP vs the fp32 oracle (the realistic inputs assert it; every case dumps
    # the numbers: gpurun_out/parity_stats -> profiles/r05_gradient_anchor_stats.md)
    G.assert_gradients_anchored(G.gradient_anchor_stats(hg, og, g64), name=stats_name, all_rows_tol=all_rows_tol)
    return of


@pytest.mark.parametrize("opacity", [0.999, None], ids=["opaque", "opacity_rand"])
@pytest.mark.parametrize("flavour_name", ["wodilate", "upstream"])
def test_c2_every_view_vs_oracle(oracle_mod, opacity, flavour_name):
    """C2 as SURVEY 8d defines it: all 8 orbit views, with opacity 0.999 and with random opacities.
    `upstream` (round-5 verdict, next #2b): the SAME views through the other package's low-pass -- cov + 0.3 instead of the
    fork's max(cov, 0.3) (LoG/model/geometry.py:87-88 against LoG/cuda/compute_radius_kernel.cu:100-104), which keeps every
    2-D covariance >= 0.3 I and its conic well conditioned -- with north_star's PLAIN criterion asserted: rel-L2 <= 1e-4 over
    ALL rows of all seven gradient tensors, HIP against float64 and against the fp32 oracle.  It holds there; what the fork's
    flavour misses on these uniform draws (profiles/r05_gradient_anchor_stats.md, rows c2_*) is therefore the clamp's
    conditioning (needle splats whose clamped covariance is nearly singular), not the kernels."""
    from log_amd import rasterizer as R, scenes
    cams = scenes.orbit_cameras(8, W=1920, H=1080, focal=2139.0)
    sc = scenes.random_scene(1_000_000, seed=0, opacity=opacity)
    up = flavour_name == "upstream"
    for v, cam in enumerate(cams):
        of = _full_parity(oracle_mod, cam, sc, (1.0, 1.0, 1.0), dL_seed=1 + v, check_lists=(v in (0, 5)),
                          stats_name="c2_%s%s_view%d" % ("upstream_" if up else "", "opaque" if opacity else "rand", v),
                          flavour=R.UPSTREAM if up else R.WODILATE, all_rows_tol=1e-4 if up else None)
        assert of["I"] > 2_000_000


@pytest.mark.parametrize("n,opacity,view,W,H", [(10_000_000, None, 3, 1920, 1080), (30_000_000, 0.999, 0, 1920, 1080),
                                                (30_000_000, None, 5, 1920, 1080), (10_000_000, None, 2, 3840, 2160)],
                         ids=["10M_opacity_rand", "30M_north_star", "30M_opacity_rand", "4K_10M_opacity_rand"])
def test_full_size_vs_oracle(oracle_mod, n, opacity, view, W, H):
    """The bench workloads at full size against the oracle: 10 M (C3 scale), the 30 M north-star point with opacity 0.999
    and with random opacities (every list walked to its end), and 10 M Gaussians on the 4K tile grid of C5 -- lists (every
    tile's list = the oracle's minus provably invisible entries, same order), image / final_T / fork maps bit for bit,
    every gradient of the reverse walk, the chain rule on identical inputs, and end to end against the float64 twin."""
    cam, sc = _scene(n, W, H, opacity=opacity, view=view)
    of = _full_parity(oracle_mod, cam, sc, (1.0, 1.0, 1.0),
                      stats_name="full_%dM_%s_%dx%d" % (n // 1_000_000, "opaque" if opacity else "rand", W, H))
    assert of["I"] > n


def test_tree_ordered_heavy_tailed_vs_oracle(oracle_mod):
    """What LoG actually hands to the rasterizer (C3): the level-of-detail selection of a 10 M-point tree -- siblings in
    neighbouring rows, 3-4 % of the splats above 16 px radius (the input that exercises the huge-rect paths of the
    binning stage) -- against the oracle, full size."""
    import types
    from diff_gaussian_rasterization_wodilate import GaussianRasterizer
    import gpu_util as G
    from log_amd import lod, scenes
    dev = torch.device("cuda:0")
    W, H = 1920, 1080
    tr = scenes.synth_tree(40000, 7, 4, split_prob=0.5, hole_prob=0.02, seed=0, root_scale=0.03)
    cam = scenes.orbit_cameras(8, W=W, H=H)[2]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rast = GaussianRasterizer(raster_settings=G.settings(cam, (1.0, 1.0, 1.0), dev))
    tree = types.SimpleNamespace(node_index=t(tr["node_index"]), tree=t(tr["tree"]), max_level=30, min_resolution_pixel=3.0)
    act = types.SimpleNamespace(scaling_activation=torch.exp, rotation_activation=torch.nn.functional.normalize)
    model = types.SimpleNamespace(xyz=t(tr["xyz"]), scaling=t(tr["scaling"]), rotation=t(tr["rotation"]), activation=act)
    sel = lod.traverse(tree, model, t(tr["root_index"]), rast).cpu().numpy()
    assert sel.shape[0] > 3_000_000
    rng = np.random.default_rng(5)
    q = tr["rotation"][sel]
    sc = dict(xyz=tr["xyz"][sel], scaling=np.exp(tr["scaling"][sel]).astype(np.float32),
              rotation=(q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)).astype(np.float32),
              opacity=(1.0 / (1.0 + np.exp(-(rng.standard_normal((sel.shape[0], 1)) + 1.0)))).astype(np.float32),
              colors=rng.random((sel.shape[0], 3), dtype=np.float32))
    of = _full_parity(oracle_mod, cam, sc, (1.0, 1.0, 1.0), stats_name="tree_ordered_heavy_tailed", all_rows_tol=1e-4)
    radii = of["radii"]
    assert (radii > 16).mean() > 0.01 and of["I"] > 2 * sel.shape[0]


@pytest.mark.parametrize("n,view", [(1_000_000, 3), (30_000_000, 1)], ids=["trained_like_1M", "trained_like_30M"])
def test_trained_like_scene_vs_oracle(oracle_mod, n, view):
    """Round-4 verdict, missing #4 / next #1c: a scene with the statistics of a TRAINED model (log_amd.scenes.
    trained_like_scene: log-normal scales sigma 0.5, anisotropy <= 10, sigmoid-normal opacity) at C2 size and at the 30 M
    north-star size, 1080p: forward bit for bit in both compositing forms, lists, reverse walk, chain rule -- and the
    PLAIN gradient criterion of north_star: rel-L2 <= 1e-4 over ALL rows, end to end, against float64 and against the
    fp32 oracle (no row excluded)."""
    from log_amd import scenes
    cam = scenes.orbit_cameras(8, W=1920, H=1080, focal=2139.0)[view]
    sc = scenes.trained_like_scene(n, seed=0)
    of = _full_parity(oracle_mod, cam, sc, (1.0, 1.0, 1.0), stats_name="trained_like_%dM" % (n // 1_000_000),
                      all_rows_tol=1e-4)
    # (composited somewhere: 27 % of the Gaussians at 1 M, 8 % at 30 M -- a dense cube is still mostly occluded)
    assert of["I"] > 2 * n and 0.03 < float((of["point_weight"] > 0).mean()) < 0.9


def test_c5_band_full_size_vs_oracle(oracle_mod):
    """BASELINE configs[4] on one of its 8 GPUs AT ITS SIZE against the oracle (round-4 verdict, missing #3 / next #1a):
    100 M Gaussians, 3840x2160, band 3 of 8 (tile rows [50, 67)), through lr_project_band_kernel (the default for band
    views) and the gradient path the multi-GPU step uses (row-major sink) as well as fresh gradients -- against the
    oracle's whole-view render restricted to the band (oracle.forward(tile_rows=...)): radii of all 100 M, records of the
    band's Gaussians, the band's tile lists, image / final_T / fork maps bit for bit, reverse-walk gradients <= 1e-4, the
    chain rule end to end anchored on the float64 twin.  The record array (6.4 GB) and the accumulator rows (6.4 GB) put
    every byte offset past 4 GiB under a check for the first time."""
    import gpu_util as G
    from log_amd import dist as D, rasterizer as R
    try:   # ~90 GB of host arrays (inputs, oracle records, fp32 + float64 gradients of 100 M rows): never drive a box out of memory
        avail_gb = next(int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")) / 1e6
    except (OSError, StopIteration):
        avail_gb = 1e9
    if avail_gb < 200:
        pytest.skip("needs ~90 GB of host memory for the oracle side (MemAvailable %.0f GB)" % avail_gb)
    N, W, H, bands, band = 100_000_000, 3840, 2160, 8, 3
    cam, sc = _scene(N, W, H, seed=0, opacity=0.999, view=1)
    rows = D.band_rows(band, bands, H)
    b, e = D.band_pixels(band, bands, H)
    bg = (1.0, 1.0, 1.0)
    v, of = G.oracle_forward(oracle_mod, cam, sc, bg, tile_rows=rows)
    vis = of["radii"] > 0
    assert 10_000_000 < int(vis.sum()) < 40_000_000 and of["I"] > 20_000_000
    with R.tile_rows(*rows):
        hf = G.hip_forward(cam, sc, bg, scratch_floats=16)
    assert hf["rec"].nbytes > (1 << 32)                       # records past the 4 GiB mark are among those compared
    last = np.nonzero(vis)[0][-1]
    assert 64 * int(last) > (1 << 32)
    st = G.compare_forward(hf, of)
    for k in ("radii_mismatch", "rec_bits_mismatch", "offsets_mismatch", "list_mismatch", "n_contrib_mismatch",
              "image_bits_mismatch", "final_T_bits_mismatch", "pid_mismatch"):
        assert st[k] == 0, (k, st)
    assert st["pwp_max_abs"] == 0.0 and st["pw_max_abs"] == 0.0
    assert (hf["image"][:, :b] == 1.0).all() and (hf["image"][:, e:] == 1.0).all()     # outside the band: background
    dL = np.zeros((3, H, W), np.float32)
    dL[:, b:e] = np.random.default_rng(2).random((3, e - b, W), dtype=np.float32)
    hg = G.hip_backward(hf, dL)                               # fresh gradients: five 100 M-row tensors
    og = oracle_mod.backward(v, of, dL)
    for k in ("means2D", "conic", "opacities", "colors"):     # the reverse walk: every row
        assert rel_l2(hg[k], og[k]) < 1e-4, (k, rel_l2(hg[k], og[k]))
    # end to end against the float64 twin, on the band's rows (the statistics of 100 M mostly-zero rows would need ~40 GB
    # of float64 temporaries); every row outside the band must be exactly zero
    g64 = oracle_mod.backward_f64(v, of, dL)
    for k in ("means3D", "scales", "rotations", "means2D", "opacities", "colors"):
        assert not hg[k][~vis].any(), k
    take = lambda d: {k: (a[vis] if isinstance(a, np.ndarray) and a.shape[:1] == (N,) else a) for k, a in d.items()}
    g64v = take({k: a for k, a in g64.items() if k != "chain32"})
    g64v["chain32"] = take(g64["chain32"])
    G.assert_gradients_anchored(G.gradient_anchor_stats(take(hg), take(og), g64v), name="c5_band_100M_3840x2160")
    # the multi-GPU step's form: the same view's gradients ADDED into one 64-byte row per Gaussian (6.4 GB of running
    # sums: LOGRAST_BWD_ACCUMULATE_ROWS) -- twice, so that the second pass really adds
    rs, flavour, use_filter, m, s_, r_, saved = hf["_torch"]
    del hf["rec"], st
    dev = m.device
    sink = {"rows": torch.zeros(N, 16, device=dev)}
    g = torch.tensor(dL, device=dev)
    for rep in range(2):
        with R.tile_rows(*rows):
            hf2 = G.hip_forward(cam, sc, bg, scratch_floats=16)
        R._backend.backward(rs, flavour, use_filter, m, s_, r_, hf2["_torch"][6], g, sink=sink)
        del hf2
    torch.cuda.synchronize()
    got = sink["rows"].cpu().numpy()
    assert not got[~vis].any() and not got[:, 14:].any()
    for k, (c0, c1) in D.ROW_COLUMNS.items():
        want = 2.0 * og[k].reshape(N, -1)[vis]
        tol = 1e-4 if k in ("opacities", "colors") else 2e-3      # (chain-rule outputs: summation-order noise on a few ill-conditioned rows, as in test_band_projection_at_scale)
        assert rel_l2(got[vis][:, c0:c1], want) < tol, (k, rel_l2(got[vis][:, c0:c1], want))


def test_every_long_list_needs_its_tail(oracle_mod):
    """The worst case of the lazily ordered lists (include/lograst.h: lograst_ordered_lengths), at a size where hundreds of
    workgroups take the rare path at once: 3 M faint Gaussians on a 320x240 image -- every tile's list holds 10-20 K keys and
    no pixel ever stops, so every compositing wave parks at the end of the first window, the second sort pass orders every
    tail (skipping the window that is in place) and every wave resumes.  Against the oracle, both compositing forms, bit for
    bit, and the reverse walk's gradients."""
    import gpu_util as G
    from log_amd import scenes
    W, H = 320, 240
    cam = scenes.orbit_cameras(8, W=W, H=H, focal=2139.0 * W / 1920.0)[1]
    sc = scenes.random_scene(3_000_000, seed=4, opacity=0.02)
    bg = (0.2, 0.5, 0.8)
    v, of = G.oracle_forward(oracle_mod, cam, sc, bg)
    lens = np.diff(of["tile_offsets"].astype(np.int64))
    dL = np.random.default_rng(9).random(of["image"].shape, dtype=np.float32)
    og = oracle_mod.backward(v, of, dL)
    for form in ("rows", "quadrant"):
        hf = G.hip_forward(cam, sc, bg, scratch_floats=16, fwd_form=form)
        hl = np.diff(hf["tile_offsets"].astype(np.int64))
        long_lists = int((hl > 7680).sum())
        assert long_lists > 50, (long_lists, hl.max())        # (the cloud covers a quarter of the image)
        assert hf["lazy_lists"] < long_lists // 10, (hf["lazy_lists"], long_lists)     # (nearly) every one was finished by the second pass
        assert hf["n_contrib"].max() > 7680
        _assert_forward(hf, of, True, form)
        # the reverse walk of the same form: its visits come from hit masks that BOTH compositing passes wrote (the first up
        # to the park position, the second from there on), hundreds of chunks deep
        hg = G.hip_backward(hf, dL, bwd_form=form)
        assert hg["bwd_masks"], form
        for k in ("means2D", "conic", "opacities", "colors"):
            assert rel_l2(hg[k], og[k]) < 1e-4, (form, k, rel_l2(hg[k], og[k]))


@pytest.mark.parametrize("n,W,H", [(2_000_000, 3840, 2160), (10_000_000, 1920, 1080)],
                         ids=["c5_tile_grid_4k_2M", "c3_scale_10M_1080p"])
def test_properties_at_scale(n, W, H):
    import gpu_util as G
    from log_amd import rasterizer as R
    dev = torch.device("cuda:0")
    cam, sc = _scene(n, W, H, opacity=None, view=3)
    hf = G.hip_forward(cam, sc, (0.5, 0.5, 0.5))
    rs, flavour, use_filter, m, s, r, saved = hf["_torch"]
    ok, I = _tile_sorted_ok(saved, W, H, dev)
    assert ok and I > n                                # every tile list sorted by (depth, id)
    n_inst, over, _, n_rect = R.last_state_info()
    assert n_inst == I and not over
    assert _rects_total(saved) == n_rect >= I          # instance conservation: sum of rect areas == the rect-rule count
    assert int((saved["radii"] > 0).sum()) > 0.9 * n
    img = hf["image"]
    assert np.isfinite(img).all() and img.min() >= 0.0 and img.max() <= 1.0 + 1e-5
    fT = hf["final_T"]
    assert fT.max() <= 1.0 and fT.min() >= 1e-4 * 0.009    # T stops below 1e-4 only by one last factor >= 0.01
    offs = hf["tile_offsets"].astype(np.int64)
    gx = (W + 15) // 16
    ty, tx = np.divmod(np.arange(len(offs) - 1), gx)
    ncp = np.zeros(((H + 15) // 16 * 16, gx * 16), np.int64)
    ncp[:H, :W] = hf["n_contrib"]
    tmax = ncp.reshape(-1, 16, gx, 16).max(axis=(1, 3)).reshape(-1)
    assert (tmax <= np.diff(offs)).all()               # n_contrib never exceeds the tile's list length
    assert (hf["point_id_pixel"][hf["n_contrib"] == 0] == -1).all()
    # determinism: second run is bit-identical
    hf2 = G.hip_forward(cam, sc, (0.5, 0.5, 0.5))
    assert (hf2["image"].view(np.uint32) == img.view(np.uint32)).all()
    assert (hf2["point_list"] == hf["point_list"]).all()
    del hf2
    # the binning-stage support cull changes no output: without it every rect tile is an instance, same image bits
    prev = R.set_tile_cull(False)
    try:
        hf3 = G.hip_forward(cam, sc, (0.5, 0.5, 0.5))
    finally:
        R.set_tile_cull(prev)
    assert hf3["I"] == n_rect
    assert (hf3["image"].view(np.uint32) == img.view(np.uint32)).all()
    assert (hf3["point_id_pixel"] == hf["point_id_pixel"]).all()
    assert (hf3["point_weight"].view(np.uint32) == hf["point_weight"].view(np.uint32)).all()
    del hf3
    # backward: zero in -> zero out; homogeneous of degree 1 in dL
    dL = np.random.default_rng(2).random(img.shape, dtype=np.float32)
    g1 = G.hip_backward(hf, dL)
    g2 = G.hip_backward(hf, 2.0 * dL)
    g0 = G.hip_backward(hf, np.zeros_like(dL))
    for k in ("means2D", "conic", "opacities", "colors"):
        assert np.abs(g0[k]).max() == 0, k
        assert rel_l2(g2[k], 2.0 * g1[k]) < 1e-5, (k, rel_l2(g2[k], 2.0 * g1[k]))
        assert np.isfinite(g1[k]).all(), k


@pytest.mark.parametrize("world", [2, 3])
def test_image_split_into_tile_row_bands(oracle_mod, world):
    """SURVEY 8e second axis ("per-GPU tile ownership"), on one GPU: rendering the bands of tile rows one after the
    other (log_amd.rasterizer.tile_rows, as each rank of a node would) gives the single-GPU image bit for bit, the
    same arg-max map, radii / point_weight as the max over bands, and gradients that sum to the single-GPU ones."""
    import math
    from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer
    from log_amd import dist as D, rasterizer as R, scenes
    dev = torch.device("cuda:0")
    N, W, H = 200_000, 1280, 720
    sc = scenes.random_scene(N, seed=7, opacity=None, smax=0.02)
    cam = scenes.orbit_cameras(4, W=W, H=H, focal=1400.0)[1]
    T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam["FoVx"] * 0.5), tanfovy=math.tan(cam["FoVy"] * 0.5),
        bg=T([0.2, 0.4, 0.6]), scale_modifier=1.0, viewmatrix=T(cam["world_view_transform"]),
        projmatrix=T(cam["full_proj_transform"]), sh_degree=0, campos=T(cam["camera_center"]), prefiltered=False,
        debug=False)
    rast = GaussianRasterizer(raster_settings=rs)
    w = torch.rand(3, H, W, device=dev)
    keys = ("xyz", "scaling", "rotation", "opacity", "colors")

    def render(rows=None):
        leaves = {k: T(sc[k]).requires_grad_(True) for k in keys}
        m2 = torch.zeros(N, 3, device=dev, requires_grad=True)
        ctx = R.tile_rows(*rows) if rows else R.tile_rows(0, 0)
        with ctx:
            out = rast(means3D=leaves["xyz"], means2D=m2, shs=None, colors_precomp=leaves["colors"],
                       opacities=leaves["opacity"], scales=leaves["scaling"], rotations=leaves["rotation"],
                       cov3D_precomp=None)
        (out[0] * w).sum().backward() if rows is None else (out[0][:, rows[0] * 16:rows[1] * 16] * w[:, rows[0] * 16:rows[1] * 16]).sum().backward()
        info = R.last_state_info()
        return out, {**{k: leaves[k].grad for k in keys}, "means2D": m2.grad}, (info[0], info[3])

    full, g_full, (i_full, rect_full) = render()
    image = torch.empty_like(full[0])
    pid = torch.empty_like(full[2])
    radii = torch.zeros_like(full[1])
    pw = torch.zeros_like(full[4])
    g_sum = {k: torch.zeros_like(v) for k, v in g_full.items()}
    inst = 0
    for r in range(world):
        rows = D.band_rows(r, world, H)
        b, e = D.band_pixels(r, world, H)
        out, g, (n_inst, _) = render(rows)
        image[:, b:e] = out[0][:, b:e]
        pid[b:e] = out[2][b:e]
        assert torch.equal(out[0][:, :b], torch.tensor([0.2, 0.4, 0.6], device=dev)[:, None, None].expand(3, b, W))
        radii = torch.maximum(radii, out[1])
        pw = torch.maximum(pw, out[4])
        inst += n_inst
        for k in g_sum:
            g_sum[k] += g[k]
    assert torch.equal(image, full[0]) and torch.equal(pid, full[2])
    assert torch.equal(radii, full[1]) and torch.equal(pw, full[4])
    # a (Gaussian, tile) instance belongs to exactly one band; the count can exceed the single-GPU one by a few: a rect
    # clipped to ONE tile is not support-tested (lr_project_one), so a tile the full render culls may survive -- it
    # contributes nothing either way (the image above is bit-identical)
    assert i_full <= inst <= rect_full
    # outputs of the reverse walk: every row well conditioned -> rel-L2
    for k in ("means2D", "opacity", "colors"):
        err = float((g_sum[k] - g_full[k]).norm() / g_full[k].norm())
        assert err < 1e-5, (k, err)
    # behind the per-Gaussian chain rule: the bands' sum and the single render are two fp32 summation orders of the same
    # gradient -- on EVERY row they differ by no more than the summation-order noise the row's conditioning allows
    # (tests/gpu_util.py: ROW_FLOOR units of amplified round-off each, conditioning from the float64 twin)
    import gpu_util as G
    v64, of = G.oracle_forward(oracle_mod, cam, sc, (0.2, 0.4, 0.6))
    g64 = oracle_mod.backward_f64(v64, of, w.cpu().numpy())
    for j, (k, k64) in enumerate((("xyz", "means3D"), ("scaling", "scales"), ("rotation", "rotations"))):
        d = (g_sum[k] - g_full[k]).reshape(N, -1).double().norm(dim=1).cpu().numpy()
        unit, well = G.row_units(g64, k64, j)
        assert (d <= 2.0 * G.ROW_FLOOR * unit).all(), (k, float((d / np.maximum(unit, 1e-300))[unit > 0].max()))
        assert rel_l2(g_full[k].cpu().numpy().astype(np.float64)[well], g64[k64][well]) < 1e-4, k


@pytest.mark.parametrize("world", [3, 8])
def test_band_prepass_selects_exactly_the_gaussians_the_band_keeps(world):
    """"every GPU preprocesses only Gaussians whose rect intersects its band".  lograst_tile_rows gives
    every Gaussian's tile-row range from the projection's own code; for every band of `world`: the selected set is exactly
    the set a full-input forward clipped to the band keeps (radii > 0), and rendering FROM THE SUBSET gives that band bit
    for bit (image, arg-max map through the index, point_weight) with gradients equal to the full-input render's rows."""
    import math
    from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer
    from log_amd import dist as D, rasterizer as R, scenes
    dev = torch.device("cuda:0")
    N, W, H = 150_000, 1280, 720
    sc = scenes.random_scene(N, seed=9, opacity=None, smax=0.03)
    sc["xyz"][:2000] *= 4.0                                   # some Gaussians outside the frustum / behind the camera
    cam = scenes.orbit_cameras(4, W=W, H=H, focal=1400.0)[2]
    T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam["FoVx"] * 0.5), tanfovy=math.tan(cam["FoVy"] * 0.5),
        bg=T([0.2, 0.4, 0.6]), scale_modifier=1.0, viewmatrix=T(cam["world_view_transform"]),
        projmatrix=T(cam["full_proj_transform"]), sh_degree=0, campos=T(cam["camera_center"]), prefiltered=False,
        debug=False)
    rast = GaussianRasterizer(raster_settings=rs)
    w = torch.rand(3, H, W, device=dev)
    keys = ("xyz", "scaling", "rotation", "opacity", "colors")
    base = {k: T(sc[k]) for k in keys}
    y0, y1 = rast.tile_rows(base["xyz"], base["scaling"], base["rotation"])
    gy = (H + 15) // 16
    assert int(y1.max()) <= gy and bool(((y1 > y0) | ((y0 == 0) & (y1 == 0))).all())

    def render(index, rows):
        leaves = {k: (base[k] if index is None else base[k][index]).clone().requires_grad_(True) for k in keys}
        n = leaves["xyz"].shape[0]
        m2 = torch.zeros(n, 3, device=dev, requires_grad=True)
        with R.tile_rows(*rows):
            out = rast(means3D=leaves["xyz"], means2D=m2, shs=None, colors_precomp=leaves["colors"],
                       opacities=leaves["opacity"], scales=leaves["scaling"], rotations=leaves["rotation"],
                       cov3D_precomp=None)
        b, e = rows[0] * 16, min(rows[1] * 16, H)
        (out[0][:, b:e] * w[:, b:e]).sum().backward()
        return out, {**{k: leaves[k].grad for k in keys}, "means2D": m2.grad}

    total = 0
    for r in range(world):
        rows = D.band_rows(r, world, H)
        b, e = D.band_pixels(r, world, H)
        idx = D.band_index(y0, y1, r, world, H)
        full, g_full = render(None, rows)
        assert torch.equal(idx, torch.nonzero(full[1] > 0).reshape(-1)), r      # exactly what the clipped projection keeps
        total += int(idx.numel())
        sub, g_sub = render(idx, rows)
        assert torch.equal(sub[0][:, b:e], full[0][:, b:e])
        assert torch.equal(sub[1], full[1][idx]) and torch.equal(sub[4], full[4][idx])
        pid = sub[2][b:e]
        assert torch.equal(torch.where(pid >= 0, idx[pid.clamp(min=0).long()], pid.long()), full[2][b:e].long())
        for k in ("means2D", "opacity", "colors"):
            ref = g_full[k][idx]
            assert float((g_sub[k] - ref).norm()) <= 1e-5 * float(ref.norm()) + 1e-12, (r, k)
            rest = torch.ones(N, dtype=torch.bool, device=dev)
            rest[idx] = False
            assert float(g_full[k][rest].abs().sum()) == 0.0, (r, k)               # nobody else has a gradient in this band
    assert total >= int(((y1 > y0)).sum())                                        # bands overlap where a rect straddles them


@pytest.mark.parametrize("n,bands,band", [(4_000_000, 8, 3), (4_000_000, 8, 0), (3_000_001, 2, 1), (1_000_000, 27, 13)])
def test_band_projection_at_scale(n, bands, band):
    """lr_project_band_kernel against the full-view kernel on the same band of a 3840x2160 view (LOGRAST_BAND_SPARSE 1 / 0):
    several projection workgroups with several batches each, rings that wrap, a band of one plane (half the image), a band
    of five tile rows, a Gaussian count that ends inside a wave -- radii, tile lists, image, fork maps bit for bit,
    records of the Gaussians that have a rect bit for bit, gradients to summation-order noise."""
    import gpu_util as G
    from log_amd import dist as D, rasterizer as R, tune
    W, H = 3840, 2160
    cam, sc = _scene(n, W, H, seed=3, opacity=None, view=2)
    rows = D.band_rows(band, bands, H)
    b, e = D.band_pixels(band, bands, H)
    dL = np.zeros((3, H, W), np.float32)
    dL[:, b:e] = np.random.default_rng(2).random((3, e - b, W), dtype=np.float32)
    res = {}
    tune.reset_knobs()
    try:
        for sparse in (0, 1):
            tune.set_knob("LOGRAST_BAND_SPARSE", sparse)
            with R.tile_rows(*rows):
                hf = G.hip_forward(cam, sc, (0.1, 0.2, 0.3), scratch_floats=16)
                res[sparse] = (hf, G.hip_backward(hf, dL))
    finally:
        tune.reset_knobs()
    (a, ga), (s, gs) = res[0], res[1]
    vis = a["radii"] > 0
    assert 0.001 < vis.mean() < 0.9 and a["I"] > 10000, (vis.mean(), a["I"])
    for k in ("image", "final_T", "point_weight_pixel", "point_weight"):
        assert (a[k].view(np.uint32) == s[k].view(np.uint32)).all(), k
    for k in ("radii", "tile_offsets", "point_list", "n_contrib", "point_id_pixel"):
        assert (a[k] == s[k]).all(), k
    assert (a["rec"][vis].view(np.uint32) == s["rec"][vis].view(np.uint32)).all()
    # the two runs share every kernel behind the projection: their gradients differ by the order of the reverse walk's
    # float atomics only -- 1e-5 on its own outputs; behind the chain rule a few ill-conditioned rows (near-isotropic
    # Gaussians: the rotation gradient is a difference of nearly equal terms) carry that noise into the L2 norm at 2e-4
    # (same kernel, two runs; the row-wise criteria of tests/gpu_util.py are what bounds those rows against float64)
    for k in ga:
        assert rel_l2(gs[k], ga[k]) < (1e-5 if k in ("means2D", "conic", "opacities", "colors") else 2e-3), k
        assert np.abs(gs[k][~vis]).max() == 0 or k == "conic", k

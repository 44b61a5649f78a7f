"""Boundary behaviour the reference's callers rely on: input layouts, streams, interleaved
forwards, degenerate sizes and values.  Every case is also checked against the oracle where it applies."""
import numpy as np
import pytest
import torch

from util import rel_l2, small_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _leaves(sc, dev=DEV, dtype=torch.float32):
    T = lambda a: torch.tensor(a, device=dev, dtype=dtype, requires_grad=True)
    return dict(means3D=T(sc["xyz"]), scales=T(sc["scaling"]), rotations=T(sc["rotation"]),
                opacities=T(sc["opacity"]), colors_precomp=T(sc["colors"]))


def _call(rast, lv, m2=None, **kw):
    if m2 is None:
        m2 = torch.zeros_like(lv["means3D"], dtype=torch.float32, requires_grad=True)
    return rast(means2D=m2, shs=None, cov3D_precomp=None, **lv, **kw), m2


def test_noncontiguous_and_float64_inputs_match_contiguous():
    from diff_gaussian_rasterization_wodilate import GaussianRasterizer
    import gpu_util as G
    cam, sc = small_case(n=800, W=80, H=64, focal=90.0, seed=21, smax=0.1)
    rast = GaussianRasterizer(raster_settings=G.settings(cam, (0.2, 0.3, 0.4), torch.device(DEV)))
    ref, _ = _call(rast, _leaves(sc))
    # results of cat / slicing / transposes, as LoG's get_all produces them (level_of_gaussian.py:262-296)
    wide = {k: torch.tensor(np.concatenate([v, v], 1), device=DEV) for k, v in sc.items()}
    lv = dict(means3D=wide["xyz"][:, :3], scales=wide["scaling"][:, 3:], rotations=wide["rotation"][:, 4:],
              opacities=wide["opacity"][:, :1], colors_precomp=wide["colors"].t().contiguous().t()[:, :3])
    assert not lv["means3D"].is_contiguous()
    out, _ = _call(rast, lv)
    assert torch.equal(out[0], ref[0]) and torch.equal(out[2], ref[2])
    out64, _ = _call(rast, _leaves(sc, dtype=torch.float64))
    assert torch.equal(out64[0], ref[0])
    # opacities as [N] instead of [N,1]
    lv1 = _leaves(sc)
    lv1["opacities"] = lv1["opacities"].detach().reshape(-1).requires_grad_(True)
    out1, _ = _call(rast, lv1)
    out1[0].sum().backward()
    assert torch.equal(out1[0], ref[0]) and lv1["opacities"].grad.shape == (800,)


def test_side_stream_and_interleaved_forwards(oracle_mod):
    """LoG runs forward #1 (render_to_check, no_grad), forward #2, then backward #2 on the same stream
    (level_of_gaussian.py:207-241, renderer.py:153); forward #1 must not clobber what backward #2 needs."""
    from diff_gaussian_rasterization_wodilate import GaussianRasterizer
    import gpu_util as G
    camA, scA = small_case(n=1200, W=96, H=72, focal=100.0, seed=22, smax=0.08)
    camB, scB = small_case(n=900, W=96, H=72, focal=100.0, seed=23, smax=0.12, view=2)
    dev = torch.device(DEV)
    w = torch.tensor(np.random.default_rng(5).random((3, 72, 96), dtype=np.float32), device=dev)

    def grads(stream, interleave):
        with torch.cuda.stream(stream):
            rastB = GaussianRasterizer(raster_settings=G.settings(camB, (0, 0, 0), dev))
            rastA = GaussianRasterizer(raster_settings=G.settings(camA, (0, 0, 0), dev))
            lv = _leaves(scA)
            out, m2 = _call(rastA, lv)
            if interleave:
                with torch.no_grad():
                    _call(rastB, {k: v.detach() for k, v in _leaves(scB).items()})
                    _call(rastA, {k: v.detach() * 1.01 for k, v in lv.items()})
            (out[0] * w).sum().backward()
        stream.synchronize()
        return out[0].detach().clone(), {k: v.grad.clone() for k, v in lv.items()}, m2.grad.clone()

    img0, g0, m0 = grads(torch.cuda.current_stream(dev), False)
    img1, g1, m1 = grads(torch.cuda.Stream(device=dev), True)
    assert torch.equal(img0, img1)
    for k in g0:
        assert rel_l2(g1[k].cpu().numpy(), g0[k].cpu().numpy()) < 1e-5, k
    assert rel_l2(m1.cpu().numpy(), m0.cpu().numpy()) < 1e-5


def test_backward_twice_with_retain_graph():
    from diff_gaussian_rasterization_wodilate import GaussianRasterizer
    import gpu_util as G
    cam, sc = small_case(n=500, W=64, H=48, focal=70.0, seed=24)
    rast = GaussianRasterizer(raster_settings=G.settings(cam, (1, 1, 1), torch.device(DEV)))
    lv = _leaves(sc)
    out, _ = _call(rast, lv)
    loss = out[0].sum()
    loss.backward(retain_graph=True)
    g1 = lv["means3D"].grad.clone()
    loss.backward()
    # atomics reorder the sums between the two passes: compare in relative L2, like every other gradient test
    assert rel_l2(lv["means3D"].grad.cpu().numpy(), (2 * g1).cpu().numpy()) < 1e-5


@pytest.mark.parametrize("W,H", [(1, 1), (15, 17), (16, 16), (33, 5)])
def test_tiny_and_ragged_images(oracle_mod, W, H):
    import gpu_util as G
    from log_amd import scenes
    cam = scenes.orbit_cameras(1, W=W, H=H, focal=max(W, H) * 1.2)[0]
    sc = scenes.random_scene(300, seed=25, opacity=None, smax=0.3)
    hf = G.hip_forward(cam, sc, (0.1, 0.5, 0.9))
    _, of = G.oracle_forward(oracle_mod, cam, sc, (0.1, 0.5, 0.9))
    st = G.compare_forward(hf, of)
    assert st["radii_mismatch"] == 0 and st["list_mismatch"] == 0 and st["image_bits_mismatch"] == 0 and \
        st["n_contrib_mismatch"] == 0 and st["pid_mismatch"] == 0, st


def test_non_finite_inputs_are_contained(oracle_mod):
    """NaN / inf positions and scales make THAT Gaussian invisible (robustness rule of the projection stage);
    the rest of the scene renders exactly as without them; no hang, no crash."""
    import gpu_util as G
    cam, sc = small_case(n=1000, W=96, H=80, focal=100.0, seed=26, smax=0.1)
    bad = {k: v.copy() for k, v in sc.items()}
    bad["xyz"][10] = np.nan
    bad["xyz"][11, 0] = np.inf
    bad["scaling"][12] = np.nan
    bad["scaling"][13, 1] = np.inf
    bad["rotation"][14] = np.nan
    hf = G.hip_forward(cam, bad, (0, 0, 0))
    _, ob = G.oracle_forward(oracle_mod, cam, bad, (0, 0, 0))
    assert (hf["radii"] == ob["radii"]).all()             # same decisions as the oracle on garbage, too
    assert (hf["radii"][10:12] == 0).all()                # non-finite positions are culled
    assert (hf["point_weight"][10:15] == 0).all()         # non-finite covariances never contribute
    keep = np.ones(1000, bool)
    keep[10:15] = False
    clean = {k: v[keep] for k, v in sc.items()}
    _, of = G.oracle_forward(oracle_mod, cam, clean, (0, 0, 0))
    assert (hf["image"].view(np.uint32) == of["image"].view(np.uint32)).all()
    assert np.isfinite(hf["image"]).all()
    hg = G.hip_backward(hf, np.ones_like(hf["image"]))
    for k in ("means3D", "scales", "rotations", "opacities", "colors"):
        assert np.isfinite(hg[k][keep]).all(), k
    assert (hg["means3D"][10:12] == 0).all()


def test_mark_visible_and_settings_fields():
    from diff_gaussian_rasterization_wodilate import GaussianRasterizer
    import gpu_util as G
    cam, sc = small_case(n=200, seed=27)
    rs = G.settings(cam, (0, 0, 0), torch.device(DEV))
    rast = GaussianRasterizer(raster_settings=rs)
    for f in ("projmatrix", "viewmatrix", "tanfovx", "tanfovy", "image_width", "image_height"):
        assert hasattr(rast.raster_settings, f)          # read at level_of_gaussian.py:73-78
    vis = rast.markVisible(torch.tensor(sc["xyz"], device=DEV))
    assert vis.dtype == torch.bool and vis.shape == (200,) and bool(vis.all())


@pytest.mark.parametrize("n", [300_000, 8_000_000], ids=["short_lists", "long_lists"])
def test_layered_depths_keep_the_bucket_sort_and_match_the_oracle(oracle_mod, n):
    """Two thin slabs perpendicular to the view direction with a gap between them (a foreground in front of a
    background): every tile's depths fall into two narrow clusters.  A linear depth -> bucket map spends its buckets on
    the gap; the sample-equalised map (sort.hip: LrDepthMap) keeps the clusters spread.  Lists and image vs the oracle."""
    import gpu_util as G
    from log_amd import scenes
    rng = np.random.default_rng(9)
    cam = scenes.orbit_cameras(8, W=1920, H=1080)[0]                   # at (3, 0, 0), looking down -x
    sc = scenes.random_scene(n, seed=9, opacity=None)
    layer = np.where(rng.random(n) < 0.5, 0.45, -0.45).astype(np.float32)
    sc["xyz"][:, 0] = layer + (rng.standard_normal(n) * 2e-3).astype(np.float32)
    hf = G.hip_forward(cam, sc, (0.0, 0.0, 0.0))
    v, of = G.oracle_forward(oracle_mod, cam, sc, (0.0, 0.0, 0.0))
    st = G.compare_forward(hf, of)
    for k in ("radii_mismatch", "offsets_mismatch", "list_mismatch", "n_contrib_mismatch", "image_bits_mismatch", "pid_mismatch"):
        assert st[k] == 0, (k, st)
    assert np.diff(of["tile_offsets"].astype(np.int64)).max() > (8192 if n > 1_000_000 else 400)

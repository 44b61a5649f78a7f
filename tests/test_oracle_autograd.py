"""The C oracle's hand-written backward (reverse walk + chain rule) against an independent float64
autograd restatement of the forward (oracle/torch_oracle.py).  Also exercises analytic invariants the
reference's consumers rely on (known-answer checks)."""
import numpy as np
import pytest
import torch

from util import cam_tan, oracle_view, rel_l2, small_case


def _torch_run(torch_oracle, cam, sc, bg, dL, filter_mode, ndc_cull):
    T = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    leaves = dict(means3D=T(sc["xyz"]), scales=T(sc["scaling"]), rotations=T(sc["rotation"]),
                  opacities=T(sc["opacity"]), colors=T(sc["colors"]))
    leaves["means2D"] = torch.zeros(len(sc["xyz"]), 3, dtype=torch.float64, requires_grad=True)
    tfx, tfy = cam_tan(cam)
    img, radii, aux = torch_oracle.render(
        cam["image_width"], cam["image_height"], tfx, tfy, torch.tensor(cam["world_view_transform"]),
        torch.tensor(cam["full_proj_transform"]), torch.tensor(bg), leaves["means3D"], leaves["means2D"],
        leaves["scales"], leaves["rotations"], leaves["opacities"], leaves["colors"],
        filter_mode=filter_mode, ndc_cull=bool(ndc_cull))
    (img * torch.tensor(dL, dtype=torch.float64)).sum().backward()
    return img.detach().numpy(), radii.numpy(), aux, {k: v.grad.numpy() for k, v in leaves.items()}


@pytest.mark.parametrize("seed,opacity,filter_mode,ndc_cull", [
    (0, None, 2, 1), (1, 0.999, 2, 1), (2, None, 1, 0), (3, None, 0, 1), (4, 0.5, 2, 0)])
def test_c_oracle_vs_float64_autograd(oracle_mod, seed, opacity, filter_mode, ndc_cull):
    from oracle import torch_oracle
    cam, sc = small_case(seed=seed, opacity=opacity)
    bg = [0.3, 0.6, 0.9]
    v = oracle_view(oracle_mod, cam, bg, filter_mode, ndc_cull)
    f = oracle_mod.forward(v, sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["colors"])
    dL = np.random.default_rng(1).random(f["image"].shape, dtype=np.float32)
    g = oracle_mod.backward(v, f, dL)
    img, radii, aux, tg = _torch_run(torch_oracle, cam, sc, bg, dL, filter_mode, ndc_cull)
    assert f["I"] > 100
    assert (radii == f["radii"]).all()
    assert rel_l2(f["image"], img) < 1e-5
    assert (aux["point_id_pixel"].numpy() == f["point_id_pixel"]).mean() > 0.995
    assert rel_l2(f["point_weight_pixel"], aux["point_weight_pixel"].numpy()) < 1e-5
    assert rel_l2(f["point_weight"], aux["point_weight"].numpy()) < 1e-5
    assert rel_l2(f["final_T"], aux["final_T"].numpy()) < 1e-5
    for k in ("means3D", "means2D", "scales", "rotations", "opacities", "colors"):
        assert rel_l2(g[k], tg[k]) < 1e-4, k


@pytest.mark.parametrize("seed,opacity,filter_mode,ndc_cull", [(0, None, 2, 1), (2, None, 1, 0), (4, 0.5, 2, 0)])
def test_float64_twin_of_the_backward_vs_float64_autograd(oracle_mod, seed, opacity, filter_mode, ndc_cull):
    """oracle.backward_f64 (the C reverse walk + chain rule in double, decisions from the fp32 forward: the anchor of the
    GPU tests' end-to-end gradient criterion) against the independent dense float64 autograd forward: an order of
    magnitude closer than the fp32 backward is (what is left is the fp32 rounding of the projected records the walk
    reads), and its per-row conditioning estimate bounds the fp32 oracle's own row errors."""
    from oracle import torch_oracle
    cam, sc = small_case(seed=seed, opacity=opacity)
    bg = [0.3, 0.6, 0.9]
    v = oracle_view(oracle_mod, cam, bg, filter_mode, ndc_cull)
    f = oracle_mod.forward(v, sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["colors"])
    dL = np.random.default_rng(1).random(f["image"].shape, dtype=np.float32)
    g32 = oracle_mod.backward(v, f, dL)
    g64 = oracle_mod.backward_f64(v, f, dL)
    _, _, _, tg = _torch_run(torch_oracle, cam, sc, bg, dL, filter_mode, ndc_cull)
    for k in ("means3D", "means2D", "scales", "rotations", "opacities", "colors"):
        assert g64[k].dtype == np.float64
        e64, e32 = rel_l2(g64[k], tg[k]), rel_l2(g32[k].astype(np.float64), tg[k])
        assert e64 < 2e-5 and e64 < max(e32, 2e-6), (k, e64, e32)
    cond = g64["cond"]
    assert cond.shape == (len(sc["xyz"]), 3) and (cond >= 0).all() and (cond[f["radii"] == 0] == 0).all()
    for j, k in enumerate(("means3D", "scales", "rotations")):
        y = np.linalg.norm(g64[k], axis=1)
        live = y > 0
        err = np.linalg.norm(g32[k].astype(np.float64) - g64[k], axis=1)[live]
        # the per-row error scale of the GPU tests (tests/gpu_util.py: amplified round-off of the inputs + the fp32
        # evaluation error of the chain rule itself): the fp32 oracle's own row errors are a few such units
        unit = (6e-8 * np.maximum(cond[:, j], 1.0) * y + np.linalg.norm(g64["chain32"][k].astype(np.float64) - g64[k], axis=1))[live]
        assert np.median(err / unit) < 5 and np.quantile(err / unit, 0.99) < 64 and (err / unit).max() < 512, \
            (k, float(np.median(err / unit)), float((err / unit).max()))


def test_invariants(oracle_mod):
    cam, sc = small_case(n=400, W=64, H=64, seed=7)
    v = oracle_view(oracle_mod, cam, (0.2, 0.4, 0.6))
    f = oracle_mod.forward(v, sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["colors"])
    pid, pwp, pw = f["point_id_pixel"], f["point_weight_pixel"], f["point_weight"]
    # image == bg where nothing contributed (LoG/render/renderer.py:157-159 strips the -1 id)
    empty = f["n_contrib"] == 0
    assert empty.any() and (pid[empty] == -1).all()
    for ch, b in enumerate((0.2, 0.4, 0.6)):
        np.testing.assert_allclose(f["image"][ch][empty], b, atol=1e-7)
    # point_weight <= min(0.99, opacity); zero for culled (LoG/model/level_of_gaussian.py:241,403)
    assert (pw <= np.minimum(sc["opacity"][:, 0], 0.99) + 1e-6).all()
    assert (pw[f["radii"] == 0] == 0).all()
    # per-pixel winner has weight <= that Gaussian's max weight, and the id indexes the input list
    ok = pid >= 0
    assert (pid[ok] < len(pw)).all()
    assert (pwp[ok] <= pw[pid[ok]] + 1e-7).all()
    assert (pwp[~ok] == 0).all()
    # tile lists are sorted by (depth, id)
    off, pl = f["tile_offsets"], f["point_list"]
    depth = f["rec"][:, 9]
    for t in range(len(off) - 1):
        ids = pl[off[t]:off[t + 1]].astype(np.int64)
        key = depth[ids].astype(np.float64) * 1e10 + ids
        assert (np.diff(key) > 0).all()


def test_opacity_zero_gives_background(oracle_mod):
    cam, sc = small_case(seed=5, opacity=0.0)
    v = oracle_view(oracle_mod, cam, (0.1, 0.2, 0.3))
    f = oracle_mod.forward(v, sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["colors"])
    np.testing.assert_allclose(f["image"][2], 0.3, atol=1e-7)
    assert (f["point_id_pixel"] == -1).all()


def test_two_gaussians_order(oracle_mod):
    """Front Gaussian (smaller view depth) must dominate regardless of input order."""
    from log_amd import scenes
    cam = scenes.orbit_cameras(1, W=32, H=32, focal=40.0, radius=3.0)[0]
    xyz = np.array([[0.5, 0, 0], [-0.5, 0, 0]], np.float32)   # camera sits at +x: first one is in front
    sca = np.full((2, 3), 0.3, np.float32)
    rot = np.tile(np.array([[1, 0, 0, 0]], np.float32), (2, 1))
    opa = np.full((2, 1), 0.95, np.float32)
    col = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    for perm in ([0, 1], [1, 0]):
        v = oracle_view(oracle_mod, cam, (0, 0, 0))
        f = oracle_mod.forward(v, xyz[perm], sca[perm], rot[perm], opa[perm], col[perm])
        c = f["image"][:, 16, 16]
        assert c[0] > 0.9 and c[1] < 0.06
        assert f["point_id_pixel"][16, 16] == perm.index(0)


def test_empty_inputs(oracle_mod):
    cam, _ = small_case()
    v = oracle_view(oracle_mod, cam, (0.5, 0.5, 0.5))
    z = lambda *s: np.zeros(s, np.float32)
    f = oracle_mod.forward(v, z(0, 3), z(0, 3), z(0, 4), z(0, 1), z(0, 3))
    assert f["I"] == 0 and (f["image"] == 0.5).all() and (f["n_contrib"] == 0).all()
    g = oracle_mod.backward(v, f, np.ones_like(f["image"]))
    assert g["means3D"].shape == (0, 3)


def _cov6(sc):
    """The covariances the scales / rotations of a scene stand for, in float64 (xx, xy, xz, yy, yz, zz)."""
    from oracle import torch_oracle
    R = torch_oracle._rot(torch.tensor(sc["rotation"], dtype=torch.float64))
    M = R * torch.tensor(sc["scaling"], dtype=torch.float64)[:, None, :]
    S = (M @ M.transpose(1, 2)).numpy()
    return np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], axis=1)


@pytest.mark.parametrize("seed,filter_mode", [(0, 2), (2, 1)])
def test_cov3d_precomp_vs_float64_autograd(oracle_mod, seed, filter_mode):
    """The rasterizer's `cov3D_precomp` input (third-party forward; LoG never passes it, renderer.py:134,149): the C
    oracle's forward from [N, 6] covariances and its dL/dcov3D (off-diagonal entries carry both symmetric positions)
    against float64 autograd with the covariance as the leaf; and the same image as the scales / rotations path when the
    covariances are the ones those stand for."""
    from oracle import torch_oracle
    cam, sc = small_case(seed=seed)
    bg = [0.3, 0.6, 0.9]
    v = oracle_view(oracle_mod, cam, bg, filter_mode, 1)
    cov = _cov6(sc).astype(np.float32)
    f = oracle_mod.forward(v, sc["xyz"], None, None, sc["opacity"], sc["colors"], cov3d=cov)
    f_sr = oracle_mod.forward(v, sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["colors"])
    assert f["I"] > 100 and rel_l2(f["image"], f_sr["image"]) < 1e-5 and (f["radii"] != f_sr["radii"]).mean() < 0.01
    dL = np.random.default_rng(1).random(f["image"].shape, dtype=np.float32)
    g = oracle_mod.backward(v, f, dL)
    assert g["cov3D"].shape == (len(cov), 6) and float(np.abs(g["scales"]).max()) == 0.0
    T = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    leaves = dict(means3D=T(sc["xyz"]), cov=T(cov), opacities=T(sc["opacity"]), colors=T(sc["colors"]))
    m2 = torch.zeros(len(cov), 3, dtype=torch.float64, requires_grad=True)
    tfx, tfy = cam_tan(cam)
    img, radii, _ = torch_oracle.render(
        cam["image_width"], cam["image_height"], tfx, tfy, torch.tensor(cam["world_view_transform"]),
        torch.tensor(cam["full_proj_transform"]), torch.tensor(bg), leaves["means3D"], m2, None, None,
        leaves["opacities"], leaves["colors"], filter_mode=filter_mode, ndc_cull=True, cov3D_precomp=leaves["cov"])
    (img * torch.tensor(dL, dtype=torch.float64)).sum().backward()
    assert (radii.numpy() == f["radii"]).all() and rel_l2(f["image"], img.detach().numpy()) < 1e-5
    assert rel_l2(g["cov3D"], leaves["cov"].grad.numpy()) < 1e-4
    assert rel_l2(g["means3D"], leaves["means3D"].grad.numpy()) < 1e-4
    assert rel_l2(g["opacities"], leaves["opacities"].grad.numpy()) < 1e-4
    assert rel_l2(g["means2D"], m2.grad.numpy()) < 1e-4
    # scale_modifier scales the `scales` only: a precomputed covariance is taken as is
    v2 = oracle_view(oracle_mod, cam, bg, filter_mode, 1)
    v2.scale_modifier = 0.5
    f2 = oracle_mod.forward(v2, sc["xyz"], None, None, sc["opacity"], sc["colors"], cov3d=cov)
    assert np.array_equal(f2["image"], f["image"])


def test_module_cov3d_precomp_through_the_test_double(oracle_mod):
    """log_amd.rasterizer's autograd surface with `cov3D_precomp` (CPU, oracle-backed test double): gradient lands on the
    covariance leaf, none on scales / rotations, and equals the oracle's own backward."""
    import oracle_backend
    from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer
    cam, sc = small_case(seed=6)
    cov = _cov6(sc).astype(np.float32)
    tfx, tfy = cam_tan(cam)
    t = lambda a: torch.tensor(np.asarray(a, np.float32))
    rs = GaussianRasterizationSettings(
        image_height=cam["image_height"], image_width=cam["image_width"], tanfovx=tfx, tanfovy=tfy, bg=t([0.1, 0.2, 0.3]),
        scale_modifier=1.0, viewmatrix=t(cam["world_view_transform"]), projmatrix=t(cam["full_proj_transform"]),
        sh_degree=0, campos=t(cam["camera_center"]), prefiltered=False, debug=False)
    old = oracle_backend.install(oracle_backend.OracleBackend())
    try:
        L = lambda a: torch.tensor(np.asarray(a, np.float32), requires_grad=True)
        m3, op, col, cv = L(sc["xyz"]), L(sc["opacity"]), L(sc["colors"]), L(cov)
        m2 = torch.zeros_like(m3, requires_grad=True)
        out = GaussianRasterizer(raster_settings=rs)(means3D=m3, means2D=m2, shs=None, colors_precomp=col, opacities=op,
                                                     scales=None, rotations=None, cov3D_precomp=cv)
        w = torch.rand(out[0].shape, generator=torch.Generator().manual_seed(0))
        out[0].backward(gradient=w)
    finally:
        oracle_backend.install(old)
    v = oracle_view(oracle_mod, cam, [0.1, 0.2, 0.3], 2, 1)
    f = oracle_mod.forward(v, sc["xyz"], None, None, sc["opacity"], sc["colors"], cov3d=cov)
    g = oracle_mod.backward(v, f, w.numpy())
    assert np.array_equal(out[0].detach().numpy(), f["image"])
    # (the oracle's reverse walk adds with float atomics under OpenMP: two runs agree to rounding, not bit for bit)
    for leaf, name in ((cv, "cov3D"), (m3, "means3D"), (op, "opacities"), (m2, "means2D"), (col, "colors")):
        assert rel_l2(leaf.grad.numpy(), g[name]) < 1e-5, name

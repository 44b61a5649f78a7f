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

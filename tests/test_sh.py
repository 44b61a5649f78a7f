"""Row N2: the packages' native `shs=` input.  CPU: the C oracle's SH basis against the reference's own PyTorch
twin (LoG/model/sh_utils.py, imported from /root/reference when present) and its hand-written backward against
float64 autograd.  GPU: the HIP kernels against the oracle, stand-alone and through the rasterizer module."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from util import cam_tan, rel_l2, small_case

REF = os.environ.get("LOG_REFERENCE", "/root/reference")


def _inputs(n=4000, M=16, seed=0, scale=0.6):
    rng = np.random.default_rng(seed)
    means = (rng.random((n, 3), dtype=np.float32) - 0.5) * 2
    campos = np.array([3.0, 0.5, -0.2], np.float32)
    shs = ((rng.random((n, M, 3), dtype=np.float32) - 0.5) * scale).astype(np.float32)
    return means, campos, shs


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "LoG")), reason="reference tree not present")
@pytest.mark.parametrize("degree", [1, 2, 3])
def test_oracle_basis_matches_reference_sh_utils(oracle_mod, degree):
    """0.5 + C0*sh0 + eval_sh_wobase(dirs, sh[1:]) == the oracle's unclamped colour (activation.py:27-34)."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from LoG.model.sh_utils import SH2RGB, eval_sh_wobase          # reference code
    means, campos, shs = _inputs(scale=0.2)                        # small coefficients: nothing reaches the clamp
    col, clamped = oracle_mod.sh_forward(means, campos, shs, degree)
    assert not clamped.any()
    d = torch.tensor(means - campos[None])
    d = d / d.norm(dim=-1, keepdim=True)
    ref = SH2RGB(torch.tensor(shs[:, 0])) + eval_sh_wobase(d, torch.tensor(shs[:, 1:]), degree=degree)
    np.testing.assert_allclose(col, ref.numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
def test_oracle_sh_backward_vs_autograd(oracle_mod, degree):
    from oracle import torch_oracle
    means, campos, shs = _inputs(n=600, seed=1, scale=2.0)         # large coefficients: the clamp is exercised
    col, clamped = oracle_mod.sh_forward(means, campos, shs, degree)
    assert degree == 0 or clamped.any()
    g = np.random.default_rng(2).random(col.shape, dtype=np.float32)
    g_shs, g_means = oracle_mod.sh_backward(means, campos, shs, degree, clamped, g)
    tm = torch.tensor(means, dtype=torch.float64, requires_grad=True)
    ts = torch.tensor(shs, dtype=torch.float64, requires_grad=True)
    out = torch_oracle.sh_colors(tm, torch.tensor(campos, dtype=torch.float64), ts, degree)
    assert rel_l2(col, out.detach().numpy()) < 1e-6
    (out * torch.tensor(g, dtype=torch.float64)).sum().backward()
    assert rel_l2(g_shs, ts.grad.numpy()) < 1e-5
    tmg = tm.grad.numpy() if tm.grad is not None else np.zeros_like(means)   # degree 0 has no view dependence
    assert np.abs(g_means - tmg).max() <= 1e-4 * max(np.abs(tmg).max(), 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("degree,M", [(0, 1), (1, 4), (2, 9), (3, 16), (1, 16)])
def test_hip_sh_kernels_vs_oracle(oracle_mod, degree, M):
    from log_amd import rasterizer as R
    dev = torch.device("cuda:0")
    means, campos, shs = _inputs(n=50_000, M=M, seed=3, scale=2.0)
    t = lambda a: torch.tensor(a, device=dev)
    col, cl = R._backend.sh_forward(t(means), t(campos), t(shs), degree)
    ocol, ocl = oracle_mod.sh_forward(means, campos, shs, degree)
    assert (cl.cpu().numpy() == ocl).all()
    assert (col.cpu().numpy().view(np.uint32) == ocol.view(np.uint32)).all()       # same op sequence: bit-exact
    g = np.random.default_rng(4).random(ocol.shape, dtype=np.float32)
    gm = torch.zeros(len(means), 3, device=dev)
    gs = R._backend.sh_backward(t(means), t(campos), t(shs), degree, cl, t(g), gm)
    ogs, ogm = oracle_mod.sh_backward(means, campos, shs, degree, ocl, g)
    assert rel_l2(gs.cpu().numpy(), ogs) < 1e-6 and rel_l2(gm.cpu().numpy(), ogm) < 1e-5


@pytest.mark.gpu
def test_module_shs_input_matches_colors_precomp_path(oracle_mod):
    """rasterizer(shs=...) == rasterizer(colors_precomp=SH colours) in image and every shared gradient, and the
    extra gradients (shs, direction term in means3D) match the oracle."""
    from diff_gaussian_rasterization import GaussianRasterizer           # upstream flavour: the one with an SH path
    import gpu_util as G
    cam, sc = small_case(n=1500, W=96, H=80, focal=100.0, seed=5, smax=0.08)
    dev = torch.device("cuda:0")
    rs = G.settings(cam, (0, 0, 0), dev)._replace(sh_degree=2)
    rng = np.random.default_rng(6)
    shs_np = ((rng.random((1500, 16, 3), dtype=np.float32) - 0.5) * 1.5).astype(np.float32)
    T = lambda a: torch.tensor(a, device=dev, requires_grad=True)
    m3, sca, rot, op, shs = T(sc["xyz"]), T(sc["scaling"]), T(sc["rotation"]), T(sc["opacity"]), T(shs_np)
    m2 = torch.zeros_like(m3, requires_grad=True)
    rast = GaussianRasterizer(raster_settings=rs)
    img, radii = rast(means3D=m3, means2D=m2, shs=shs, colors_precomp=None, opacities=op, scales=sca, rotations=rot)
    w = torch.tensor(rng.random((3, 80, 96), dtype=np.float32), device=dev)
    (img * w).sum().backward()
    # reference route: colours from the oracle, then the colors_precomp path
    ocol, ocl = oracle_mod.sh_forward(sc["xyz"], cam["camera_center"], shs_np, 2)
    m3b, scab, rotb, opb, colb = T(sc["xyz"]), T(sc["scaling"]), T(sc["rotation"]), T(sc["opacity"]), T(ocol)
    m2b = torch.zeros_like(m3b, requires_grad=True)
    img_b, _ = rast(means3D=m3b, means2D=m2b, shs=None, colors_precomp=colb, opacities=opb, scales=scab,
                    rotations=rotb)
    (img_b * w).sum().backward()
    assert torch.equal(img, img_b)
    for a, b in ((sca, scab), (rot, rotb), (op, opb), (m2, m2b)):
        assert torch.allclose(a.grad, b.grad, rtol=1e-4, atol=1e-6)
    ogs, ogm = oracle_mod.sh_backward(sc["xyz"], cam["camera_center"], shs_np, 2, ocl, colb.grad.cpu().numpy())
    assert rel_l2(shs.grad.cpu().numpy(), ogs) < 1e-4
    assert rel_l2((m3.grad - m3b.grad).cpu().numpy(), ogm) < 1e-3


@pytest.mark.gpu
def test_shs_gradients_through_the_running_sum_sink():
    """accumulate_grads_into with an "shs" entry: two views add dL/dshs (and every other attribute) straight into the
    step's running sums; same numbers as summing the per-view autograd gradients."""
    from diff_gaussian_rasterization_wodilate import GaussianRasterizer
    from log_amd import rasterizer as R
    import gpu_util as G
    cam, sc = small_case(n=1200, W=80, H=64, focal=90.0, seed=9, smax=0.08)
    cam2, _ = small_case(n=1200, W=80, H=64, focal=90.0, seed=9, smax=0.08, view=2)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(8)
    shs_np = ((rng.random((1200, 16, 3), dtype=np.float32) - 0.5) * 1.5).astype(np.float32)
    w = torch.tensor(rng.random((3, 64, 80), dtype=np.float32), device=dev)
    T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev, requires_grad=True)

    def leaves():
        return dict(means3D=T(sc["xyz"]), scales=T(sc["scaling"]), rotations=T(sc["rotation"]), opacities=T(sc["opacity"]),
                    shs=T(shs_np))

    def run(lv, sink):
        for c in (cam, cam2):
            rs = G.settings(c, (0.1, 0.2, 0.3), dev)._replace(sh_degree=3)
            m2 = torch.zeros(1200, 3, device=dev, requires_grad=True)
            out = GaussianRasterizer(raster_settings=rs)(means3D=lv["means3D"], means2D=m2, shs=lv["shs"],
                                                         colors_precomp=None, opacities=lv["opacities"],
                                                         scales=lv["scales"], rotations=lv["rotations"])
            if sink is None:
                out[0].backward(gradient=w)
            else:
                with R.accumulate_grads_into(sink):
                    out[0].backward(gradient=w)

    a = leaves()
    run(a, None)
    b = leaves()
    sink = {k: torch.zeros_like(v) for k, v in b.items()}
    run(b, sink)
    for k in a:
        assert b[k].grad is None
        assert rel_l2(sink[k].cpu().numpy(), a[k].grad.cpu().numpy()) < 1e-5, k


@pytest.mark.gpu
def test_native_shs_degree3_full_size_through_the_gradient_sink(oracle_mod):
    """The packages' native `shs=` input at C2 size: 1 M Gaussians with 16 SH coefficients each (degree 3), 1920x1080,
    5-tuple flavour, gradients added by the backward kernels straight into running sums (accumulate_grads_into with an
    "shs" entry) -- against the oracle end to end: SH colours and clamp mask bit for bit, image bit for bit, dL/dshs and
    every other attribute's gradient within 1e-4 relative L2 (means3D: the projection chain + the view-direction term)."""
    from diff_gaussian_rasterization_wodilate import GaussianRasterizer
    from log_amd import rasterizer as R, scenes
    import gpu_util as G
    dev = torch.device("cuda:0")
    N, W, H = 1_000_000, 1920, 1080
    cam = scenes.orbit_cameras(8, W=W, H=H, focal=2139.0)[3]
    sc = scenes.random_scene(N, seed=0, opacity=None)
    rng = np.random.default_rng(21)
    shs_np = ((rng.random((N, 16, 3), dtype=np.float32) - 0.5) * 1.2).astype(np.float32)
    shs_np[:, 0] += 0.8                                             # mostly positive colours, some clamped at 0
    bg = (0.1, 0.2, 0.3)
    rs = G.settings(cam, bg, dev)._replace(sh_degree=3)
    T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev, requires_grad=True)
    lv = dict(means3D=T(sc["xyz"]), scales=T(sc["scaling"]), rotations=T(sc["rotation"]), opacities=T(sc["opacity"]),
              shs=T(shs_np))
    sink = {k: torch.zeros_like(v) for k, v in lv.items()}
    m2 = torch.zeros(N, 3, device=dev, requires_grad=True)
    w_np = np.random.default_rng(22).random((3, H, W), dtype=np.float32)
    out = GaussianRasterizer(raster_settings=rs)(means3D=lv["means3D"], means2D=m2, shs=lv["shs"], colors_precomp=None,
                                                 opacities=lv["opacities"], scales=lv["scales"], rotations=lv["rotations"])
    with R.accumulate_grads_into(sink):
        out[0].backward(gradient=torch.tensor(w_np, device=dev))
    torch.cuda.synchronize()
    # the oracle's route: SH colours, then the colors_precomp path, then back through the SH polynomial
    ocol, ocl = oracle_mod.sh_forward(sc["xyz"], cam["camera_center"], shs_np, 3)
    assert 0.001 < float((ocl != 0).mean()) < 0.5
    sc_o = dict(sc, colors=ocol)
    v, of = G.oracle_forward(oracle_mod, cam, sc_o, bg)
    assert (out[0].detach().cpu().numpy().view(np.uint32) == of["image"].view(np.uint32)).all()
    assert (out[1].cpu().numpy() == of["radii"]).all() and (out[2].cpu().numpy() == of["point_id_pixel"]).all()
    og = oracle_mod.backward(v, of, w_np)
    ogs, ogm = oracle_mod.sh_backward(sc["xyz"], cam["camera_center"], shs_np, 3, ocl, og["colors"])
    assert rel_l2(sink["shs"].cpu().numpy(), ogs) < 1e-4
    assert rel_l2(sink["opacities"].cpu().numpy().reshape(-1, 1), og["opacities"]) < 1e-4
    assert rel_l2(m2.grad.cpu().numpy(), og["means2D"]) < 1e-4
    assert rel_l2(sink["means3D"].cpu().numpy(), og["means3D"] + ogm) < 1e-4
    # scales / rotations: every row against the float64 twin (tests/gpu_util.py)
    g64 = oracle_mod.backward_f64(v, of, w_np)
    for j, k in ((1, "scales"), (2, "rotations")):
        hk, ref = sink[k].cpu().numpy().astype(np.float64), g64[k]
        unit, well = G.row_units(g64, k, j)
        live = np.linalg.norm(ref, axis=1) > 0
        assert float((live & ~well).sum()) <= 0.06 * float(live.sum())
        assert rel_l2(hk[well], ref[well]) < 1e-4, k
        eh = np.linalg.norm(hk - ref, axis=1)
        eo = np.linalg.norm(og[k].astype(np.float64) - ref, axis=1)
        assert (eh <= 2.0 * eo + G.ROW_FLOOR * unit).all(), (k, float(((eh - 2.0 * eo) / np.maximum(unit, 1e-300)).max()))

"""BASELINE config C1 ("plumbing, no GPU"): the reference's UNMODIFIED Python -- LoG/render/renderer.py
(NaiveRendererAndLoss), LoG/model/base_gaussian.py, LoG/model/level_of_gaussian.py -- imported from
/root/reference and driven through this repo's drop-in packages.  On a CPU-only machine the HIP backend is
replaced by the oracle test double (tests/oracle_backend.py); everything above the rasterizer boundary is the
reference's own code.  Skipped where /root/reference does not exist (the GPU box)."""
import math
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = os.environ.get("LOG_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "LoG")), reason="reference tree not present")


@pytest.fixture()
def log_env(oracle_mod):
    """sys.path + stubs for the reference's host-only deps that are absent here (cv2) + the LoG.cuda drop-in."""
    from log_amd import rasterizer as R
    from log_amd.compute_radius import compute_radius_module
    import oracle_backend
    from oracle_backend import OracleBackend
    added = []
    if REF not in sys.path:
        sys.path.insert(0, REF)
        added.append(REF)
    stubs = {}
    if "cv2" not in sys.modules:
        stubs["cv2"] = types.ModuleType("cv2")          # only used by visualisation helpers
    drop = types.ModuleType("LoG.cuda.compute_radius")  # what INTEGRATION.md installs as LoG/cuda/compute_radius.py
    drop.compute_radius_module = compute_radius_module
    stubs["LoG.cuda.compute_radius"] = drop
    sys.modules.update(stubs)
    old = oracle_backend.install(OracleBackend())
    yield
    oracle_backend.install(None if isinstance(old, R.HipBackend) else old)
    for k in stubs:
        sys.modules.pop(k, None)
    for p in added:
        sys.path.remove(p)


def _batch(cams):
    keys = ["camera_center", "world_view_transform", "full_proj_transform", "K", "R", "T"]
    cam = {k: torch.tensor(np.stack([c[k] for c in cams])) for k in keys}
    for k in ("image_width", "image_height", "FoVx", "FoVy"):
        cam[k] = [c[k] for c in cams]
    return {"camera": cam}


@pytest.mark.parametrize("n,W,focal,smax", [(1500, 96, 110.0, 0.06), (50000, 400, 445.0, None)],
                         ids=["small", "C1_50k_400x400_2views"])
def test_reference_renderer_runs_unchanged_on_dropin(log_env, oracle_mod, n, W, focal, smax):
    """ids[1] is BASELINE.json configs[0] (C1, "plumbing, no GPU"): 50 000 Gaussians, 400x400, 2 views, through
    NaiveRendererAndLoss + BaseGaussian.create_from_record."""
    from LoG.render.renderer import NaiveRendererAndLoss          # reference code, unmodified
    from LoG.model.base_gaussian import BaseGaussian              # reference code, unmodified
    from log_amd import scenes
    H = W
    cams = scenes.orbit_cameras(2, W=W, H=H, focal=focal)
    sc = scenes.random_scene(n, seed=0, opacity=None, smax=smax)
    sc["opacity"] = np.clip(sc["opacity"], 0.05, 0.95)
    model = BaseGaussian.create_from_record({k: v for k, v in sc.items()})
    renderer = NaiveRendererAndLoss(split="train", use_origin_render=False, background=[1., 1., 1.])
    batch = _batch(cams)
    batch["image"] = torch.rand(2, H, W, 3)
    model.train()
    out = renderer(batch, model)                                   # vis() -> render() -> rasterizer(**name_args)
    assert out["render"].shape == (2, 3, H, W)
    assert len(out["point_id"]) == 2 and out["point_id"][0].dtype in (torch.int32, torch.int64)
    assert out["radii"][0].shape == (n,) and out["point_weight"][0].shape == (n,)
    out["loss"].backward()
    for name in ("xyz", "colors", "scaling", "opacity", "rotation"):
        g = getattr(model, name).grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0, name
    vs = out["viewspace_points"][0]
    assert vs.grad is not None and float(vs.grad[:, :2].abs().sum()) > 0      # consumed by counter.py:40,46
    # the image is exactly what the oracle renders for the activated parameters the reference handed over
    ret = model.get_all()
    tfx, tfy = math.tan(cams[0]["FoVx"] * 0.5), math.tan(cams[0]["FoVy"] * 0.5)
    v = oracle_mod.make_view(W, H, tfx, tfy, cams[0]["world_view_transform"], cams[0]["full_proj_transform"], [1, 1, 1])
    f = oracle_mod.forward(v, ret["xyz"].detach().numpy(), ret["scaling"].detach().numpy(),
                           ret["rotation"].detach().numpy(), ret["opacity"].detach().numpy(),
                           ret["colors"].detach().numpy())
    np.testing.assert_array_equal(out["render"][0].detach().numpy(), f["image"])
    # upstream flavour (use_origin_render=True, apps/check_gui.py:19): 2-tuple path of renderer.py:160-165
    r2 = NaiveRendererAndLoss(split="demo", use_origin_render=True, background=[0., 0., 0.])
    model.eval()
    with torch.no_grad():
        out2 = r2.vis(batch, model)
    assert out2["render"].shape == (2, 3, H, W)
    # eval mode of the fork passes use_filter=False (renderer.py:151-152)
    r3 = NaiveRendererAndLoss(split="demo", use_origin_render=False, background=[0., 0., 0.])
    with torch.no_grad():
        out3 = r3.vis(batch, model)
    assert out3["render"].shape == (2, 3, H, W)


def test_reference_lod_compute_radius_call_site(log_env, oracle_mod):
    """LoG/model/level_of_gaussian.py:65-88 (Gaussian.compute_radius) calls compute_radius_module with the
    rasterizer's settings; the module object installed as LoG.cuda.compute_radius is ours."""
    from LoG.model.level_of_gaussian import Gaussian               # reference code, unmodified
    from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer
    from log_amd import scenes
    cam = scenes.orbit_cameras(1, W=128, H=96, focal=140.0)[0]
    sc = scenes.random_scene(500, seed=2, smax=0.05)
    g = Gaussian()
    g.xyz = torch.tensor(sc["xyz"])
    g.scaling = torch.log(torch.tensor(sc["scaling"]))
    g.rotation = torch.tensor(sc["rotation"])
    tfx, tfy = math.tan(cam["FoVx"] * 0.5), math.tan(cam["FoVy"] * 0.5)
    rs = GaussianRasterizationSettings(
        image_height=96, image_width=128, tanfovx=tfx, tanfovy=tfy, bg=torch.zeros(3), scale_modifier=1.0,
        viewmatrix=torch.tensor(cam["world_view_transform"]), projmatrix=torch.tensor(cam["full_proj_transform"]),
        sh_degree=0, campos=torch.tensor(cam["camera_center"]), prefiltered=False, debug=False)
    rast = GaussianRasterizer(raster_settings=rs)
    index = torch.arange(0, 500, 2)
    scaling3d, radius2d = g.compute_radius(index, rast)
    assert radius2d.shape == (250,) and scaling3d.shape == (250,)
    ref = oracle_mod.compute_radius(sc["xyz"][::2], sc["scaling"][::2], sc["rotation"][::2],
                                    cam["full_proj_transform"], cam["world_view_transform"],
                                    128 / (2 * tfx), 96 / (2 * tfy), tfx, tfy)
    np.testing.assert_allclose(radius2d.numpy(), ref, rtol=1e-5, atol=1e-5)
    # fork-only method used at level_of_gaussian.py:59
    r2 = rast.compute_radius(g.xyz, torch.tensor(sc["scaling"]), g.rotation)
    np.testing.assert_allclose(r2.numpy()[::2], ref, rtol=1e-5, atol=1e-5)


class _Cfg(dict):
    """Attribute + item access, like the reference's yacs-style config nodes."""
    __getattr__ = dict.__getitem__


def _log_model(seed, n):
    """The reference's LoG model (level_of_gaussian.py:174-196) on a synthetic cloud, with a two-level tree grown
    through its own densification plumbing (tree.split_and_remove + Splitter, as update_depth_stage :510-515 does)."""
    from LoG.model.level_of_gaussian import LoG
    from log_amd import scenes
    sc = scenes.random_scene(n, seed=seed, smax=0.08)
    cfg_opt = _Cfg(optimize_keys=["xyz", "colors", "scaling", "opacity", "rotation", "shs"], opt_all_levels=True,
                   lr_dict={"xyz": 0.00016, "xyz_final": 0.0000016, "colors": 0.0025, "shs": 0.000125, "scaling": 0.005,
                            "opacity": 0.05, "rotation": 0.001, "max_steps": 300})
    torch.manual_seed(seed)
    model = LoG(gaussian=_Cfg(init_ply=dict(filename={"xyz": sc["xyz"], "colors": sc["colors"]}, scale3d=1.0,
                                            init_opacity=0.3), sh_degree=1, xyz_scale=1.0),
                tree=_Cfg(max_child=4, max_level=30), optimizer=cfg_opt, densify_and_remove=_Cfg())
    model.base_iter = 1
    model.set_stage("tree")
    model.training_setup()
    model.upgrade_tree()
    gen = torch.Generator().manual_seed(seed + 1)
    for level in range(2):
        leaf = (model.tree.node_index == -1) & (model.tree.depth == level)
        flag_split = leaf & (torch.rand(leaf.shape[0], generator=gen) < 0.5)
        flag_remove = torch.zeros_like(flag_split)
        flag_split, flag_remove = model.tree.split_and_remove(flag_split, flag_remove)
        model.splitter.split_and_remove(model.gaussian, model.optimizer, flag_split, flag_remove, remove_split=False)
        model.splitter.split_and_remove_other(model.counter, ["create_steps", "radius3d_min", "radius3d_max"],
                                              flag_split, flag_remove, remove_split=False)
        model.counter.reset(model.num_points)
    model.counter.radius3d_max.fill_(10.0)
    model.counter.radius3d_min.fill_(1e-4)
    model.gaussian.active_sh_degree = 1          # exercise the shs key (activation.py:27-34)
    model.train()
    return model


def _run_steps(model, steps, W, H):
    """trainer.py:144-160 (training_step): render -> loss.backward -> update_by_output -> step."""
    from LoG.render.renderer import NaiveRendererAndLoss
    from log_amd import scenes
    renderer = NaiveRendererAndLoss(split="train", use_origin_render=False, background=[1., 1., 1.])
    cams = scenes.orbit_cameras(steps, W=W, H=H, focal=1.1 * W, radius=2.2)
    torch.manual_seed(11)
    selected = []
    for it in range(steps):
        batch = _batch([cams[it]])
        batch["image"] = torch.rand(1, H, W, 3)
        output = renderer(batch, model)
        selected.append(torch.cat([output["visibility_flag"][0]["index"], output["visibility_flag"][0]["index_node"]]).clone())
        output["loss"].backward()
        model.update_by_output(output)
        model.step()
    return selected


@pytest.fixture()
def cpu_cuda_shims(log_env):
    """create_from_point (LoG/utils/file.py:88-91) calls distCUDA2(xyz.cuda()): on the CPU-only machine `.cuda()` is
    made the identity and the 3-NN distance comes from scipy (TEST DOUBLES for host-only plumbing)."""
    from scipy.spatial import cKDTree

    def dist2(points):
        p = points.detach().cpu().numpy().astype(np.float64)
        d, _ = cKDTree(p).query(p, k=4)
        return torch.from_numpy((d[:, 1:] ** 2).mean(axis=1).astype(np.float32))

    mod = types.ModuleType("simple_knn._C")
    mod.distCUDA2 = dist2
    old_mod, old_cuda = sys.modules.get("simple_knn._C"), torch.Tensor.cuda
    sys.modules["simple_knn._C"] = mod
    torch.Tensor.cuda = lambda self, *a, **k: self
    yield
    torch.Tensor.cuda = old_cuda
    if old_mod is not None:
        sys.modules["simple_knn._C"] = old_mod
    else:
        sys.modules.pop("simple_knn._C", None)


def test_reference_training_step_with_all_dropins_installed(cpu_cuda_shims):
    """The reference's own LoG model, tree, counter, optimizer and renderer run three training steps twice: as they
    are, and with the N2/N3/N4 drop-ins installed on their classes (log_amd.{lod,counter,sparse_optimizer,get_all}.install()).
    Same selected points in the same order, same integer counters, parameters and Adam moments within fp32
    round-off (the arithmetic below the boundary is the oracle's in both runs)."""
    from LoG.model.tensor_tree import TensorTree
    from LoG.model.counter import Counter
    from LoG.model.sparse_optimizer import SparseOptimizer
    from LoG.model.level_of_gaussian import LoG
    from log_amd import lod, counter, sparse_optimizer, get_all
    saved = (TensorTree.traverse, Counter.update_by_output, SparseOptimizer.step, SparseOptimizer.load_state_dict)
    saved_get_all = LoG.get_all
    import LoG.render.renderer as ref_renderer
    saved_torch = ref_renderer.torch
    W, H = 96, 72
    try:
        ref = _log_model(0, 400)
        assert ref.tree.num_nodes > 50 and int(ref.tree.depth.max()) == 2
        sel_ref = _run_steps(ref, 3, W, H)
        import log_amd
        patched = log_amd.install_all()                 # = the four install() calls of INTEGRATION.md 3b
        assert [c.__name__ for c in patched] == ["LoG", "TensorTree", "Counter", "SparseOptimizer"]
        # ... and the torch.unique call inside renderer.py (:156) now goes through log_amd.counter.torch_unique, with
        # torch's own result (leading -1 included), while every other use of `torch` in that module is torch's
        assert ref_renderer.torch is not torch and ref_renderer.torch.unique is counter.torch_unique
        assert ref_renderer.torch.zeros_like is torch.zeros_like and ref_renderer.torch.float32 is torch.float32
        pid_map = torch.tensor([[3, -1, 3], [0, 7, -1]], dtype=torch.int32)
        for tagged in (False, True):
            if tagged:
                pid_map._lograst_num_gaussians = 9
            got = ref_renderer.torch.unique(pid_map, sorted=True, return_counts=True)
            want = torch.unique(pid_map, sorted=True, return_counts=True)
            assert all(torch.equal(a, b) and a.dtype == b.dtype for a, b in zip(got, want)), tagged
        # a map without an empty pixel: the stand-in keeps the leading -1 with a ZERO count (no read-back per view to
        # decide; renderer.py:157-159 strips the entry), everything behind it is torch's result
        full = torch.tensor([[3, 1, 3], [0, 7, 7]], dtype=torch.int32)
        full._lograst_num_gaussians = 9
        gi, gc = ref_renderer.torch.unique(full, sorted=True, return_counts=True)
        wi, wc = torch.unique(full, sorted=True, return_counts=True)
        assert int(gi[0]) == -1 and int(gc[0]) == 0 and torch.equal(gi[1:], wi) and torch.equal(gc[1:], wc)
        new = _log_model(0, 400)
        sel_new = _run_steps(new, 3, W, H)
    finally:
        ref_renderer.torch = saved_torch
        TensorTree.traverse, Counter.update_by_output, SparseOptimizer.step, SparseOptimizer.load_state_dict = saved
        LoG.get_all = saved_get_all
        if hasattr(SparseOptimizer, "_lograst_load_state_dict"):
            del SparseOptimizer._lograst_load_state_dict
    for a, b in zip(sel_ref, sel_new):
        assert a.numel() > 100 and torch.equal(a, b)
    for k in ("radii_max", "visible_count", "radii_max_max", "area_sum", "create_steps"):
        assert torch.equal(getattr(ref.counter, k), getattr(new.counter, k)), k
    assert int(ref.counter.area_sum.sum()) > 0 and int(ref.counter.visible_count.max()) >= 2
    for k in ("weights_max", "weights_sum", "grad_sum"):
        torch.testing.assert_close(getattr(new.counter, k), getattr(ref.counter, k), rtol=1e-4, atol=1e-6)
    assert float(new.optimizer.global_steps) == float(ref.optimizer.global_steps) == 3.0
    moved = 0.0
    cfg_lr = {"colors": 0.0025, "shs": 0.000125, "opacity": 0.05, "rotation": 0.001}
    for k in ("xyz", "colors", "scaling", "opacity", "rotation", "shs"):
        p, q = getattr(new.gaussian, k), getattr(ref.gaussian, k)
        # Adam turns a round-off-sized gradient into a full +-lr step (m / sqrt(v) = +-1 whatever the magnitude; e.g.
        # the real part of the identity quaternions here, whose analytic gradient is 0), so a handful of elements
        # may differ by up to 2 * lr per step; everything else agrees to fp32 round-off.
        lr = {"xyz": 0.00016, "scaling": 0.005}.get(k, cfg_lr.get(k))
        d = (p - q).abs()
        assert float((d > 1e-5 * (1 + q.abs())).float().mean()) < 0.02, k
        assert float(d.max()) <= 2.5 * lr * 3, k
        a, b = new.optimizer.exp_avg[k], ref.optimizer.exp_avg[k]       # sums of gradients: compared in rel-L2
        assert float(b.norm()) > 0 and float((a - b).norm() / b.norm()) < 2e-3, k   # three chained steps: step 1 round-off feeds steps 2 and 3
        moved += float(ref.optimizer.exp_avg[k].abs().sum())
    assert moved > 0

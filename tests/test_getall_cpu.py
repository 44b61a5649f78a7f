"""Rows N2/N3: the fused LoG.get_all + activate_root_return drop-in.  CPU side: host logic and the
oracle's restatement (through the test double) against activations and autograd gradients produced by the
reference's own Activation class (tests/golden/make_golden_getall.py)."""
import os

import numpy as np
import pytest
import torch

import getall_util as U


@pytest.fixture()
def double(oracle_mod):
    from log_amd import rasterizer as R
    import oracle_backend
    from oracle_backend import OracleBackend
    old = oracle_backend.install(OracleBackend())
    yield
    oracle_backend.install(None if isinstance(old, R.HipBackend) else old)


@pytest.mark.parametrize("path", U.GOLDEN, ids=[os.path.basename(p) for p in U.GOLDEN])
def test_get_all_matches_reference_activation_and_autograd(path, double):
    from log_amd import get_all
    g = np.load(path)
    model, camera = U.log_like(g, "cpu")
    U.check(g, model, camera, get_all.get_all)


def test_eval_mode_and_unfixed_parent(double):
    from log_amd import get_all
    g = np.load(U.GOLDEN[1])
    model, camera = U.log_like(g, "cpu", training=False)
    ret = get_all.get_all(model, camera, None)
    params = model.gaussian.visibility_flag["params"]
    assert not any(isinstance(p, torch.nn.Parameter) for p in params.values())
    assert not any(v.requires_grad for v in ret.values())
    np.testing.assert_allclose(ret["colors"].numpy(), g["act_colors"], rtol=3e-6, atol=1e-6)
    # fix_parent=False: node rows are parameters too (level_of_gaussian.py:282-293)
    model, camera = U.log_like(g, "cpu", fix_parent=False)
    ret = get_all.get_all(model, camera, None)
    n_all = g["index"].shape[0] + g["index_node"].shape[0]
    assert all(p.shape[0] == n_all for p in model.gaussian.visibility_flag["params"].values())
    ret["scaling"].sum().backward()
    gs = model.gaussian.visibility_flag["params"]["scaling"].grad
    np.testing.assert_allclose(gs.numpy(), ret["scaling"].detach().numpy(), rtol=1e-6)      # d exp = exp


def test_nothing_selected_and_bad_models(double):
    from log_amd import get_all
    g = np.load(U.GOLDEN[0])
    model, camera = U.log_like(g, "cpu")
    model.gaussian.visibility_flag = {"index": torch.zeros(0, dtype=torch.int64)}
    ret = get_all.get_all(model, camera, None)
    assert ret["xyz"].shape == (0, 3) and ret["opacity"].shape == (0, 1)
    model, camera = U.log_like(g, "cpu")
    model.gaussian.keys.append("extra")
    model.gaussian.extra = torch.zeros(3)
    with pytest.raises(NotImplementedError):
        get_all.get_all(model, camera, None)


def test_product_path_refuses_cpu_tensors():
    from log_amd import get_all, _lib
    g = np.load(U.GOLDEN[0])
    model, camera = U.log_like(g, "cpu")
    with pytest.raises(_lib.LograstError):
        get_all.get_all(model, camera, None)


def test_a_training_view_leaves_no_tensor_in_a_reference_cycle(double):
    """What a view allocates must die with the view's last reference, not at the next run of Python's cyclic collector:
    an autograd Function whose ctx can reach one of its own output tensors (output -> grad_fn -> ctx -> output) keeps the
    gathered rows, the AccumulateGrad nodes and through them the parameters' .grad alive -- GBs per C3 view, each a fresh
    hipMalloc (round-2 verdict, weak #7).  One get_all + rasterizer forward / backward through the drop-ins, then every
    reference dropped: the collector must find no tensor and no autograd node."""
    import gc
    import weakref
    from log_amd import get_all, rasterizer as R
    from log_amd import scenes
    from util import cam_tan
    g = np.load(U.GOLDEN[1])
    cam = scenes.orbit_cameras(2, W=64, H=48, focal=60.0)[0]
    tfx, tfy = cam_tan(cam)
    T = lambda a: torch.tensor(np.asarray(a, np.float32))
    rs = R.GaussianRasterizationSettings(48, 64, tfx, tfy, T([1, 1, 1]), 1.0, T(cam["world_view_transform"]),
                                         T(cam["full_proj_transform"]), 0, T(cam["camera_center"]), False, False)
    rast = R.GaussianRasterizer(raster_settings=rs)
    watch = []

    model, camera = U.log_like(g, "cpu")       # long-lived, like LoG's model (the stand-in is itself a cycle: keep it alive)
    flags0 = dict(model.gaussian.visibility_flag)

    def view():
        model.gaussian.visibility_flag = dict(flags0)
        act = get_all.get_all(model, camera, rast)
        m2 = torch.zeros_like(act["xyz"], requires_grad=True)
        out = rast(means3D=act["xyz"], means2D=m2, shs=None, colors_precomp=act["colors"], opacities=act["opacity"],
                   scales=act["scaling"], rotations=act["rotation"], cov3D_precomp=None)
        out[0].sum().backward()
        params = model.gaussian.visibility_flag["params"]
        assert params["scaling"].grad is not None
        watch.extend(weakref.ref(t) for t in (act["colors"], act["xyz"], params["scaling"], params["scaling"].grad, out[0], out[4]))
        model.gaussian.visibility_flag = None     # (LoG replaces it at the next view)

    gc.collect()
    gc.disable()
    try:
        view()
        assert [w() is None for w in watch] == [True] * len(watch)        # freed by reference counting alone
        gc.set_debug(gc.DEBUG_SAVEALL)
        gc.collect()
        bad = [type(o).__name__ for o in gc.garbage if isinstance(o, torch.Tensor) or type(o).__name__.endswith("Backward")]
        assert not bad, bad
    finally:
        gc.set_debug(0)
        gc.garbage.clear()
        gc.enable()

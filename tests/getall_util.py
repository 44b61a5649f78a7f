"""Shared by the CPU and GPU tests of the fused get_all drop-in: rebuild a LoG-shaped object from a golden file
(tests/golden/make_golden_getall.py) and compare the drop-in's outputs and gradients with the reference's."""
import glob
import os
import types

import numpy as np
import torch

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "getall_*.npz")))
KEYS = ["scaling", "colors", "xyz", "opacity", "rotation", "shs"]     # GaussianPoint.keys order (level_of_gaussian.py:155-160)


def log_like(g, device, training=True, fix_parent=True):
    keys = [k for k in KEYS if "model_" + k in g]
    gaussian = types.SimpleNamespace(keys=keys, active_sh_degree=int(g["degree"]))
    for k in keys:
        setattr(gaussian, k, torch.from_numpy(g["model_" + k]).to(device))
    gaussian.items = lambda: ((k, getattr(gaussian, k)) for k in keys)
    flags = {"index": torch.from_numpy(g["index"]).to(device)}
    if g["index_node"].shape[0]:
        flags["index_node"] = torch.from_numpy(g["index_node"]).to(device)
    gaussian.visibility_flag = flags
    model = types.SimpleNamespace(gaussian=gaussian, fix_parent=fix_parent, training=training)
    camera = {"camera_center": torch.from_numpy(g["camera_center"]).to(device)}
    return model, camera


def check(g, model, camera, get_all, rtol=3e-6, grad_rtol=2e-5):
    ret = get_all(model, camera, None)
    assert list(ret) == ["xyz", "scaling", "opacity", "rotation", "colors"]
    for k in ret:
        np.testing.assert_allclose(ret[k].detach().cpu().numpy(), g["act_" + k], rtol=rtol, atol=1e-6, err_msg=k)
    params = model.gaussian.visibility_flag["params"]
    assert list(params) == model.gaussian.keys
    n_leaf = g["index"].shape[0]
    for k, p in params.items():
        assert isinstance(p, torch.nn.Parameter) and p.shape[0] == n_leaf
        np.testing.assert_array_equal(p.detach().cpu().numpy(), g["model_" + k][g["index"]])
    loss = sum((ret[k] * torch.from_numpy(g["up_" + k]).to(ret[k].device)).sum() for k in ret)
    loss.backward()
    for k, p in params.items():
        if not int(g["has_grad_" + k]):
            assert p.grad is None, k
            continue
        want = g["grad_" + k]
        got = p.grad.cpu().numpy()
        assert got.shape == want.shape, k
        err = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
        assert err < grad_rtol, (k, err)
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-5 * np.abs(want).max(), err_msg=k)

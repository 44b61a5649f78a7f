"""Rows N2/N3 on the GPU: lograst_gather_activate / lograst_activate_backward through the drop-in
(log_amd/get_all.py) against the reference's Activation + torch autograd (tests/golden/getall_*.npz), and at 1 M
rows against the same op sequence in torch on the device."""
import os

import numpy as np
import pytest
import torch

import getall_util as U

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("path", U.GOLDEN, ids=[os.path.basename(p) for p in U.GOLDEN])
def test_get_all_matches_reference_activation_and_autograd(path):
    from log_amd import get_all
    g = np.load(path)
    model, camera = U.log_like(g, DEV)
    U.check(g, model, camera, get_all.get_all)


def _torch_get_all(bufs, index, index_node, campos, degree):
    """level_of_gaussian.py:262-296 + activation.py:27-44 with torch ops (what LoG runs today)."""
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    params = {k: torch.nn.Parameter(v[index]) for k, v in bufs.items()}
    full = {k: torch.cat([params[k], v[index_node]]) for k, v in bufs.items()}
    colors = full["colors"] * C0 + 0.5
    if degree > 0:
        d = full["xyz"].detach() - campos[None]
        d = d / torch.norm(d, dim=-1, keepdim=True)
        x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
        sh = full["shs"]
        colors = colors + (-C1 * y * sh[:, 0] + C1 * z * sh[:, 1] - C1 * x * sh[:, 2])
    act = {"xyz": full["xyz"], "scaling": torch.exp(full["scaling"]), "opacity": torch.sigmoid(full["opacity"]),
           "rotation": torch.nn.functional.normalize(full["rotation"]), "colors": colors}
    return params, act


@pytest.mark.parametrize("degree", [0, 1])
def test_one_million_rows_against_torch_ops(degree):
    import types
    from log_amd import get_all
    P, n_leaf, n_node, K = 3_000_000, 1_000_000, 50_000, 3
    gen = torch.Generator(device=DEV).manual_seed(degree)
    bufs = {"scaling": torch.randn(P, 3, device=DEV, generator=gen) - 3, "colors": torch.randn(P, 3, device=DEV, generator=gen),
            "xyz": torch.rand(P, 3, device=DEV, generator=gen) - 0.5, "opacity": torch.randn(P, 1, device=DEV, generator=gen),
            "rotation": torch.randn(P, 4, device=DEV, generator=gen), "shs": torch.randn(P, K, 3, device=DEV, generator=gen)}
    perm = torch.randperm(P, device=DEV, generator=gen)
    index, index_node = perm[:n_leaf], perm[n_leaf:n_leaf + n_node]
    campos = torch.tensor([0.2, 2.4, -0.7], device=DEV)
    keys = list(bufs)
    gaussian = types.SimpleNamespace(keys=keys, active_sh_degree=degree, items=lambda: ((k, bufs[k]) for k in keys),
                                     visibility_flag={"index": index, "index_node": index_node})
    model = types.SimpleNamespace(gaussian=gaussian, fix_parent=True, training=True)
    ret = get_all.get_all(model, {"camera_center": campos}, None)
    params = gaussian.visibility_flag["params"]
    p_ref, a_ref = _torch_get_all(bufs, index, index_node, campos, degree)
    ups = {k: torch.randn(v.shape, device=DEV, generator=gen) for k, v in a_ref.items()}
    sum((ret[k] * ups[k]).sum() for k in ups).backward()
    sum((a_ref[k] * ups[k]).sum() for k in ups).backward()
    for k in a_ref:
        torch.testing.assert_close(ret[k], a_ref[k], rtol=3e-6, atol=1e-6, msg=k)
    for k in keys:
        if k == "shs" and degree == 0:
            assert params[k].grad is None and p_ref[k].grad is None
            continue
        assert torch.equal(params[k].detach(), p_ref[k].detach()), k
        err = float((params[k].grad - p_ref[k].grad).norm() / p_ref[k].grad.norm())
        assert err < 2e-6, (k, err)


def test_nothing_selected_on_the_device():
    """LoG.prepare can hand over an empty selection (renderer.py:119-127 copes with N = 0 downstream)."""
    import types
    from log_amd import get_all
    g = np.load(U.GOLDEN[1])
    model, camera = U.log_like(g, DEV)
    model.gaussian.visibility_flag = {"index": torch.zeros(0, dtype=torch.int64, device=DEV)}
    ret = get_all.get_all(model, camera, None)
    assert ret["xyz"].shape == (0, 3) and ret["opacity"].shape == (0, 1) and ret["rotation"].shape == (0, 4)
    sum(v.sum() for v in ret.values()).backward()
    params = model.gaussian.visibility_flag["params"]
    assert all(p.shape[0] == 0 for p in params.values())

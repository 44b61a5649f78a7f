"""Helpers shared by the CPU and GPU tests of the N4 drop-ins (counter / id histogram / sparse Adam): rebuild the
objects the drop-in methods expect (the attributes they read from LoG's Counter / SparseOptimizer) from the golden
files written by tests/golden/make_golden_train.py."""
import os
import types

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")
COUNTER_DTYPES = {"weights_max": torch.float32, "weights_sum": torch.float32, "grad_sum": torch.float32,
                  "radii_max": torch.int16, "visible_count": torch.int16, "radii_max_max": torch.int32,
                  "area_sum": torch.int32, "create_steps": torch.int32}
ADAM_KEYS = ["xyz", "colors", "scaling", "opacity", "rotation", "shs"]
LR = {"colors": 0.0025, "shs": 0.000125, "opacity": 0.05, "rotation": 0.001}


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def fresh_counter(P, device):
    """Counter.__init__ (LoG/model/counter.py:5-19): zeros of the registered dtypes."""
    return types.SimpleNamespace(**{k: torch.zeros(P, dtype=dt, device=device) for k, dt in COUNTER_DTYPES.items()})


def counter_output(g, device, with_lists=True):
    """The `output` dict Counter.update_by_output receives (LoG/render/renderer.py:168-184, lists over views)."""
    nv = int(g["n_views"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    out = {k: [] for k in ("render", "visibility_flag", "viewspace_points", "radii", "point_weight", "point_id",
                           "point_count")}
    for v in range(nv):
        vis = g[f"v{v}_visible_index"]
        n_leaf = int(0.5 * int(g["P"]))
        out["render"].append(None)
        out["visibility_flag"].append({"index": t(vis[:n_leaf]), "index_node": t(vis[n_leaf:])})
        out["viewspace_points"].append(types.SimpleNamespace(grad=t(g[f"v{v}_grad"])))
        out["radii"].append(t(g[f"v{v}_radii"]))
        out["point_weight"].append(t(g[f"v{v}_point_weight"]))
        if with_lists:
            out["point_id"].append(t(g[f"v{v}_point_id"]))
            out["point_count"].append(t(g[f"v{v}_point_count"]))
    return out


def check_counter(counter, g, rtol=2e-6):
    for k, dt in COUNTER_DTYPES.items():
        got, want = getattr(counter, k).cpu().numpy(), g["final_" + k]
        if dt.is_floating_point:
            np.testing.assert_allclose(got, want, rtol=rtol, atol=1e-12, err_msg=k)
        else:
            np.testing.assert_array_equal(got, want, err_msg=k)


def fresh_optimizer(g, device):
    """What SparseOptimizer.__init__ sets up (sparse_optimizer.py:118-160) that step() reads."""
    amsgrad = bool(int(g["amsgrad"]))
    model = types.SimpleNamespace(**{k: torch.from_numpy(g["init_" + k].copy()).to(device) for k in ADAM_KEYS})
    zeros = lambda: {k: torch.zeros_like(getattr(model, k)) for k in ADAM_KEYS}
    opt = types.SimpleNamespace(
        global_steps=torch.tensor(float(g["start_global_steps"]), dtype=torch.float32, device=device),
        lr_dict=dict(LR), exp_avg=zeros(), exp_avg_sq=zeros(), use_amsgrad=amsgrad, xyz_lr=None,
        scaling_scheduler_args=lambda step: 0.005)
    if amsgrad:
        opt.max_exp_avg_sq = zeros()
    return model, opt


def adam_step_inputs(g, it, device):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    params = {}
    for k in ADAM_KEYS:
        p = torch.nn.Parameter(t(g[f"s{it}_param_{k}"]))
        if f"s{it}_grad_{k}" in g:
            p.grad = t(g[f"s{it}_grad_{k}"])
        params[k] = p
    return t(g[f"s{it}_index"]), params, t(g[f"s{it}_flag_vis"])


def run_adam(g, device, step_fn):
    model, opt = fresh_optimizer(g, device)
    for it in range(int(g["n_steps"])):
        index, params, flag_vis = adam_step_inputs(g, it, device)
        lr_xyz = float(g[f"s{it}_lr_xyz"])
        opt.xyz_scheduler_args = lambda step, lr=lr_xyz: lr       # the schedule itself is host code of the reference
        step_fn(opt, model, index, params, flag_vis)
        assert opt.xyz_lr == lr_xyz
    return model, opt


def check_adam(model, opt, g, rtol=2e-6):
    for k in ADAM_KEYS:
        np.testing.assert_allclose(getattr(model, k).cpu().numpy(), g["final_" + k], rtol=rtol, atol=1e-9, err_msg=k)
        np.testing.assert_allclose(opt.exp_avg[k].cpu().numpy(), g["final_exp_avg_" + k], rtol=rtol, atol=1e-12, err_msg=k)
        np.testing.assert_allclose(opt.exp_avg_sq[k].cpu().numpy(), g["final_exp_avg_sq_" + k], rtol=rtol, atol=1e-20, err_msg=k)
        if opt.use_amsgrad:
            np.testing.assert_allclose(opt.max_exp_avg_sq[k].cpu().numpy(), g["final_max_exp_avg_sq_" + k], rtol=rtol,
                                       atol=1e-20, err_msg=k)
    assert float(opt.global_steps) == float(g["final_global_steps"])

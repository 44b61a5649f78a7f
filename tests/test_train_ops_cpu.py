"""N4: id histogram, Counter.update_by_output, SparseOptimizer.step.  CPU side: the oracle's
restatements and the drop-ins' host logic (through the oracle test double) against results produced by the
reference's own classes (tests/golden/make_golden_train.py); with the reference tree present, the drop-ins are
also installed on the real classes and run side by side with the unpatched code."""
import os
import sys

import numpy as np
import pytest
import torch

import train_util as U

REF = os.environ.get("LOG_REFERENCE", "/root/reference")


@pytest.fixture()
def double(oracle_mod):
    from log_amd import rasterizer as R
    import oracle_backend
    from oracle_backend import OracleBackend
    old = oracle_backend.install(OracleBackend())
    yield
    oracle_backend.install(None if isinstance(old, R.HipBackend) else old)


def test_oracle_id_histogram_matches_torch_unique(oracle_mod):
    g = U.load("counter_a.npz")
    for v in range(int(g["n_views"])):
        ids, counts = oracle_mod.id_histogram(g[f"v{v}_pid_map"])
        np.testing.assert_array_equal(ids, g[f"v{v}_point_id"])
        np.testing.assert_array_equal(counts, g[f"v{v}_point_count"])
        assert counts.sum() == (g[f"v{v}_pid_map"] >= 0).sum()


def test_oracle_counter_matches_reference_counter(oracle_mod):
    g = U.load("counter_a.npz")
    P = int(g["P"])
    state = {k: np.zeros(P, dt) for k, dt in oracle_mod.COUNTER_FIELDS}
    for v in range(int(g["n_views"])):
        flag = oracle_mod.counter_update(state, g[f"v{v}_visible_index"], g[f"v{v}_grad"], g[f"v{v}_radii"],
                                         g[f"v{v}_point_weight"], g[f"v{v}_point_id"], g[f"v{v}_point_count"])
        np.testing.assert_array_equal(flag, g[f"v{v}_flag_vis"])
    for k, dt in oracle_mod.COUNTER_FIELDS:
        if np.issubdtype(dt, np.floating):
            np.testing.assert_allclose(state[k], g["final_" + k], rtol=2e-6, atol=1e-12, err_msg=k)
        else:
            np.testing.assert_array_equal(state[k], g["final_" + k], err_msg=k)
    assert state["area_sum"].sum() > 0 and state["visible_count"].max() >= 2


def test_counter_dropin_host_logic(double):
    from log_amd import counter
    g = U.load("counter_a.npz")
    c = U.fresh_counter(int(g["P"]), "cpu")
    out = U.counter_output(g, "cpu")
    counter.update_by_output(c, out, fix_parent=True)
    U.check_counter(c, g)
    for v in range(int(g["n_views"])):
        vf = out["visibility_flag"][v]
        np.testing.assert_array_equal(vf["flag_vis"].numpy(), g[f"v{v}_flag_vis"])
        np.testing.assert_array_equal(vf["index_vis"].numpy(), np.nonzero(g[f"v{v}_flag_vis"])[0])
        ids, counts = counter.unique_ids(torch.from_numpy(g[f"v{v}_pid_map"]), int(g[f"v{v}_radii"].shape[0]))
        np.testing.assert_array_equal(ids.numpy(), g[f"v{v}_point_id"])
        np.testing.assert_array_equal(counts.numpy(), g[f"v{v}_point_count"])


@pytest.mark.parametrize("name", ["adam_a.npz", "adam_ams.npz"])
def test_adam_dropin_host_logic_matches_reference_optimizer(name, double):
    from log_amd import sparse_optimizer
    g = U.load(name)
    model, opt = U.run_adam(g, "cpu", sparse_optimizer.step)
    U.check_adam(model, opt, g)


def test_product_paths_refuse_cpu_tensors():
    from log_amd import counter, sparse_optimizer, _lib
    g = U.load("counter_a.npz")
    with pytest.raises(_lib.LograstError):
        counter.update_by_output(U.fresh_counter(int(g["P"]), "cpu"), U.counter_output(g, "cpu"))
    with pytest.raises(_lib.LograstError):
        counter.unique_ids(torch.from_numpy(g["v0_pid_map"]), 10)
    a = U.load("adam_ams.npz")
    with pytest.raises(_lib.LograstError):
        U.run_adam(a, "cpu", sparse_optimizer.step)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "LoG")), reason="reference tree not present")
def test_installed_on_reference_classes_side_by_side(double):
    """install() patches LoG's own Counter / SparseOptimizer; patched and unpatched objects see the same calls."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import types
    from LoG.model.counter import Counter
    from LoG.model.sparse_optimizer import SparseOptimizer
    from log_amd import counter, sparse_optimizer
    ref_update, ref_step, ref_load = Counter.update_by_output, SparseOptimizer.step, SparseOptimizer.load_state_dict
    g = U.load("counter_a.npz")
    a = U.load("adam_a.npz")

    def make_opt():
        model = types.SimpleNamespace(**{k: torch.from_numpy(a["init_" + k].copy()) for k in U.ADAM_KEYS})
        lr = {"xyz": 0.00016, "xyz_final": 0.0000016, "colors": 0.0025, "shs": 0.000125, "scaling": 0.005,
              "opacity": 0.05, "rotation": 0.001, "max_steps": 30000}
        opt = SparseOptimizer(list(U.ADAM_KEYS), lr, model, device=torch.device("cpu"), xyz_scale=1.0)
        opt.global_steps += 40
        return model, opt

    try:
        c_ref = Counter(num_points=int(g["P"]))
        c_ref.update_by_output(U.counter_output(g, "cpu"), fix_parent=True)
        m_ref, o_ref = make_opt()
        for it in range(int(a["n_steps"])):
            index, params, flag_vis = U.adam_step_inputs(a, it, "cpu")
            o_ref.step(m_ref, index, params, flag_vis)
        counter.install()
        sparse_optimizer.install()
        c_new = Counter(num_points=int(g["P"]))
        c_new.update_by_output(U.counter_output(g, "cpu"), fix_parent=True)
        for k in U.COUNTER_DTYPES:
            x, y = getattr(c_new, k).numpy(), getattr(c_ref, k).numpy()
            if x.dtype.kind == "f":
                np.testing.assert_allclose(x, y, rtol=2e-6, atol=1e-12, err_msg=k)
            else:
                np.testing.assert_array_equal(x, y, err_msg=k)
        m_new, o_new = make_opt()
        for it in range(int(a["n_steps"])):
            index, params, flag_vis = U.adam_step_inputs(a, it, "cpu")
            o_new.step(m_new, index, params, flag_vis)
        assert o_new.xyz_lr == o_ref.xyz_lr and float(o_new.global_steps) == float(o_ref.global_steps)
        for k in U.ADAM_KEYS:
            np.testing.assert_allclose(getattr(m_new, k).numpy(), getattr(m_ref, k).numpy(), rtol=2e-6, atol=1e-9, err_msg=k)
            np.testing.assert_allclose(o_new.exp_avg_sq[k].numpy(), o_ref.exp_avg_sq[k].numpy(), rtol=2e-6, atol=1e-20)
        # state dicts stay interchangeable, and loading one resets the host-side step mirror
        o_new.load_state_dict(o_ref.state_dict())
        assert o_new._lograst_steps is None
    finally:
        Counter.update_by_output, SparseOptimizer.step, SparseOptimizer.load_state_dict = ref_update, ref_step, ref_load
        if hasattr(SparseOptimizer, "_lograst_load_state_dict"):
            del SparseOptimizer._lograst_load_state_dict


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "LoG")), reason="reference tree not present")
@pytest.mark.parametrize("seed", range(6))
def test_random_views_and_steps_against_reference_classes(seed, double):
    """Random sizes, random subsets of keys without gradient, amsgrad on/off, several views / steps: the reference's
    Counter and SparseOptimizer (unpatched, torch CPU) and the drop-in functions see identical inputs."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import types
    from LoG.model.counter import Counter
    from LoG.model.sparse_optimizer import SparseOptimizer
    from log_amd import counter, sparse_optimizer
    rng = np.random.default_rng(100 + seed)
    g = torch.Generator().manual_seed(seed)
    P = int(rng.integers(50, 3000))
    # ---- counter: views with random visible sets, radii (some 0 = culled), id lists over the visible ones
    views = []
    for _ in range(int(rng.integers(1, 4))):
        nv = int(rng.integers(1, P + 1))
        vis = rng.permutation(P)[:nv]
        n_leaf = int(rng.integers(0, nv + 1))
        radii = rng.integers(0, 40000, nv).astype(np.int32) * (rng.random(nv) < 0.8)       # > int16 range too (.short() wraps)
        k = int(rng.integers(0, nv + 1))
        pid = np.sort(rng.permutation(nv)[:k]).astype(np.int32)
        views.append(dict(index=vis[:n_leaf], index_node=vis[n_leaf:], grad=rng.standard_normal((nv, 3)).astype(np.float32),
                          radii=radii.astype(np.int32), pw=rng.random(nv).astype(np.float32), pid=pid,
                          pc=rng.integers(1, 5000, k).astype(np.int64)))

    def output():
        t = torch.from_numpy
        return {"render": [None] * len(views),
                "visibility_flag": [{"index": t(v["index"]), "index_node": t(v["index_node"])} for v in views],
                "viewspace_points": [types.SimpleNamespace(grad=t(v["grad"])) for v in views],
                "radii": [t(v["radii"]) for v in views], "point_weight": [t(v["pw"]) for v in views],
                "point_id": [t(v["pid"]) for v in views], "point_count": [t(v["pc"]) for v in views]}

    c_ref, c_new = Counter(num_points=P), Counter(num_points=P)
    o_ref, o_new = output(), output()
    c_ref.update_by_output(o_ref, fix_parent=True)
    counter.update_by_output(c_new, o_new, fix_parent=True)
    for k in U.COUNTER_DTYPES:
        a, b = getattr(c_new, k).numpy(), getattr(c_ref, k).numpy()
        if a.dtype.kind == "f":
            np.testing.assert_allclose(a, b, rtol=3e-6, atol=1e-12, err_msg=k)
        else:
            np.testing.assert_array_equal(a, b, err_msg=k)
    for v in range(len(views)):
        assert torch.equal(o_new["visibility_flag"][v]["flag_vis"], o_ref["visibility_flag"][v]["flag_vis"])
        assert torch.equal(o_new["visibility_flag"][v]["index_vis"], o_ref["visibility_flag"][v]["index_vis"])
    # ---- sparse Adam
    amsgrad = bool(seed % 2)
    shapes = {"xyz": (3,), "colors": (3,), "scaling": (3,), "opacity": (1,), "rotation": (4,), "shs": (int(rng.integers(1, 16)), 3)}
    lr = {"xyz": 0.00016, "xyz_final": 0.0000016, "colors": 0.0025, "shs": 0.000125, "scaling": 0.005, "opacity": 0.05,
          "rotation": 0.001, "max_steps": 30000}

    def make():
        gg = torch.Generator().manual_seed(seed)
        model = types.SimpleNamespace(**{k: torch.randn(P, *s, generator=gg) for k, s in shapes.items()})
        opt = SparseOptimizer(list(shapes), dict(lr), model, device=torch.device("cpu"), xyz_scale=1.0, use_amsgrad=amsgrad)
        opt.global_steps += int(seed * 7)
        return model, opt

    (m_ref, op_ref), (m_new, op_new) = make(), make()
    for _ in range(int(rng.integers(1, 4))):
        m = int(rng.integers(1, P + 1))
        index = torch.randperm(P, generator=g)[:m]
        flag_vis = torch.rand(m, generator=g) < 0.7
        no_grad = {k for k in shapes if rng.random() < 0.25}
        grads = {k: torch.randn(m, *s, generator=g) * 10.0 ** float(rng.integers(-6, 1)) for k, s in shapes.items()}

        def params(model):
            out = {}
            for k in shapes:
                p = torch.nn.Parameter(getattr(model, k)[index].clone())
                if k not in no_grad:
                    p.grad = grads[k].clone()
                out[k] = p
            return out

        op_ref.step(m_ref, index, params(m_ref), flag_vis)
        sparse_optimizer.step(op_new, m_new, index, params(m_new), flag_vis)
    assert float(op_ref.global_steps) == float(op_new.global_steps)
    for k in shapes:
        np.testing.assert_allclose(getattr(m_new, k).numpy(), getattr(m_ref, k).numpy(), rtol=3e-6, atol=2e-8, err_msg=k)
        np.testing.assert_allclose(op_new.exp_avg[k].numpy(), op_ref.exp_avg[k].numpy(), rtol=3e-6, atol=1e-12, err_msg=k)
        np.testing.assert_allclose(op_new.exp_avg_sq[k].numpy(), op_ref.exp_avg_sq[k].numpy(), rtol=3e-6, atol=1e-20, err_msg=k)
        if amsgrad:
            np.testing.assert_allclose(op_new.max_exp_avg_sq[k].numpy(), op_ref.max_exp_avg_sq[k].numpy(), rtol=3e-6,
                                       atol=1e-20, err_msg=k)

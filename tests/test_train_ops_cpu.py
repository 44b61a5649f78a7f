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
    from oracle_backend import OracleBackend
    old = R._set_backend_for_tests(OracleBackend())
    yield
    R._set_backend_for_tests(None if isinstance(old, R.HipBackend) else old)


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

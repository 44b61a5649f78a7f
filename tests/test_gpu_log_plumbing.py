"""The reference's UNMODIFIED classes on the device (round-3 verdict, missing #2 / next #1b): `NaiveRendererAndLoss`
(LoG/render/renderer.py:117-316), `LoG` (LoG/model/level_of_gaussian.py:223-296,379-398), `TensorTree`
(tensor_tree.py:165-185), `Counter` (counter.py:36-68) and `SparseOptimizer` (sparse_optimizer.py:163-196) run three
training steps on cuda:0 after ``log_amd.install_all()`` -- every kernel below the boundary is HIP -- and the trajectory
is compared with the same three steps on the CPU, where the classes are left as they are and the rasterizer backend is the
oracle test double (tests/oracle_backend.py; what tests/test_log_plumbing_cpu.py pins).

Needs the reference tree: `LOG_REFERENCE` (default /root/reference).  It does not exist on the driver's GPU box, where this
file is skipped; `tools/run_reference_on_gpu.sh` stages a git-ignored copy for ONE gpurun call and removes it afterwards
(log kept under profiles/)."""
import math
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = os.environ.get("LOG_REFERENCE", "/root/reference")
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "LoG")), reason="reference tree not present")]

W, H, STEPS = 96, 72, 3


@pytest.fixture()
def ref_env():
    """sys.path + the stub for the reference's host-only dependency that is absent here (cv2: visualisation helpers) +
    the LoG.cuda.compute_radius module (log_amd.install_compute_radius)."""
    import log_amd
    added = []
    if REF not in sys.path:
        sys.path.insert(0, REF)
        added.append(REF)
    stub = "cv2" not in sys.modules
    if stub:
        sys.modules["cv2"] = types.ModuleType("cv2")
    log_amd.install_compute_radius()
    yield
    if stub:
        sys.modules.pop("cv2", None)
    for p in added:
        sys.path.remove(p)


def _to(obj, dev):
    if torch.is_tensor(obj):
        return obj.to(dev)
    if isinstance(obj, dict):
        return {k: _to(v, dev) for k, v in obj.items()}
    return obj


def _steps(model, dev, log):
    """trainer.py:144-160 (training_step): render -> loss.backward -> update_by_output -> step, `STEPS` times; the
    batches are generated on the CPU from fixed seeds and moved like the reference's prepare_batch (trainer.py:25-42)."""
    from LoG.render.renderer import NaiveRendererAndLoss            # reference code, unmodified
    from log_amd import scenes
    import test_log_plumbing_cpu as P
    renderer = NaiveRendererAndLoss(split="train", use_origin_render=False, background=[1., 1., 1.]).to(dev)
    cams = scenes.orbit_cameras(STEPS, W=W, H=H, focal=1.1 * W, radius=2.2)
    gen = torch.Generator().manual_seed(11)
    selected, images = [], []
    for it in range(STEPS):
        batch = P._batch([cams[it]])
        batch["image"] = torch.rand(1, H, W, 3, generator=gen)
        batch = _to(batch, dev)
        output = renderer(batch, model)
        vf = output["visibility_flag"][0]
        selected.append(torch.cat([vf["index"], vf["index_node"]]).detach().cpu().clone())
        images.append(output["render"][0].detach().cpu().clone())
        output["loss"].backward()
        model.update_by_output(output)
        model.step()
        log("step %d on %s: %d points selected, loss %.6f" % (it, dev, selected[-1].numel(), float(output["loss"])))
    return selected, images


def test_reference_classes_train_three_steps_on_the_device(ref_env, oracle_mod, capsys):
    import log_amd
    import oracle_backend
    import test_log_plumbing_cpu as P
    from log_amd import rasterizer as R
    from LoG.model.tensor_tree import TensorTree
    from LoG.model.counter import Counter
    from LoG.model.sparse_optimizer import SparseOptimizer
    from LoG.model.level_of_gaussian import LoG
    import LoG.render.renderer as ref_renderer
    lines = []

    def log(msg):
        lines.append(msg)
        with capsys.disabled():
            print("[gpu plumbing] " + msg, flush=True)

    dev = torch.device("cuda:0")
    # The reference builds its model on the host and asks the device only for the 3-NN distances
    # (base_gaussian.py:39-42: distCUDA2(xyz.cuda())): here through simple_knn._C of this repo (knn.hip).
    # (two builds from the same seeds: the construction is deterministic, as in tests/test_log_plumbing_cpu.py)
    cpu = P._log_model(0, 400)
    assert cpu.tree.num_nodes > 50 and int(cpu.tree.depth.max()) == 2
    # ---- CPU: the classes as they are, the oracle below the boundary (test double) ----
    cpu0 = {k: getattr(cpu.gaussian, k).clone() for k in ("xyz", "colors", "scaling", "opacity", "rotation", "shs")}
    cpu0 = types.SimpleNamespace(**cpu0)
    old = oracle_backend.install(oracle_backend.OracleBackend())
    try:
        sel_cpu, img_cpu = _steps(cpu, torch.device("cpu"), log)
    finally:
        oracle_backend.install(None if isinstance(old, R.HipBackend) else old)
    assert isinstance(R._backend, R.HipBackend)
    # ---- device: install_all() + HIP kernels ----
    saved = (TensorTree.traverse, Counter.update_by_output, SparseOptimizer.step, SparseOptimizer.load_state_dict,
             LoG.get_all, ref_renderer.torch)
    try:
        patched = log_amd.install_all()
        assert [c.__name__ for c in patched] == ["LoG", "TensorTree", "Counter", "SparseOptimizer"]
        gpu = P._log_model(0, 400)
        for k in ("xyz", "colors", "scaling", "opacity", "rotation", "shs"):
            assert torch.equal(getattr(gpu.gaussian, k), getattr(cpu0, k)), k
        gpu = gpu.to(dev)                              # apps/train.py:164 (trainer.to(device))
        loaded_before = _loaded_lograst()
        sel_gpu, img_gpu = _steps(gpu, dev, log)
        torch.cuda.synchronize()
    finally:
        (TensorTree.traverse, Counter.update_by_output, SparseOptimizer.step, SparseOptimizer.load_state_dict,
         LoG.get_all, ref_renderer.torch) = saved
        if hasattr(SparseOptimizer, "_lograst_load_state_dict"):
            del SparseOptimizer._lograst_load_state_dict
    assert loaded_before, "liblograst.so is not mapped into this process"
    # same points selected, in the same order, in every step (integer path)
    for it, (a, b) in enumerate(zip(sel_cpu, sel_gpu)):
        assert a.numel() > 100 and torch.equal(a, b), it
    # step 0 starts from identical parameters: the images differ only through the activations' last ulp (torch's CPU
    # exp / sigmoid / normalize against the device's) -- the image of IDENTICAL inputs is bit-identical to the oracle's,
    # which tests/test_gpu_parity.py and test_gpu_scale.py assert
    d0 = float((img_gpu[0] - img_cpu[0]).abs().max())
    log("image step 0: max |device - cpu| = %.3e" % d0)
    assert d0 < 2e-5
    for it in range(1, STEPS):
        d = float((img_gpu[it] - img_cpu[it]).abs().max())
        log("image step %d: max |device - cpu| = %.3e" % (it, d))
        assert d < 5e-2          # (Adam turns round-off-sized gradients into full steps: see the parameter check below)
    # integer counters exact, float counters / parameters / moments within fp32 round-off, as in the CPU test
    for k in ("radii_max", "visible_count", "radii_max_max", "area_sum", "create_steps"):
        assert torch.equal(getattr(cpu.counter, k), getattr(gpu.counter, k).cpu()), k
    assert int(cpu.counter.area_sum.sum()) > 0
    # (float counters accumulate per-view weights and gradient norms over the three steps: from step 1 on the two
    # trajectories' parameters differ by Adam's steps on round-off-sized gradients -- see the parameter check below -- and
    # the device's gradient sums depend on its atomics' order: nearly all elements within 1e-4, every one within 5e-3)
    for k in ("weights_max", "weights_sum", "grad_sum"):
        a, b = getattr(gpu.counter, k).cpu(), getattr(cpu.counter, k)
        off = (a - b).abs() > 1e-6 + 1e-4 * b.abs()
        log("%s: %d of %d elements beyond 1e-4, max |device - cpu| = %.3e" % (k, int(off.sum()), off.numel(), float((a - b).abs().max())))
        assert float(off.float().mean()) < 0.005, k
        torch.testing.assert_close(a, b, rtol=5e-3, atol=1e-4)
    assert float(gpu.optimizer.global_steps) == float(cpu.optimizer.global_steps) == float(STEPS)
    cfg_lr = {"xyz": 0.00016, "scaling": 0.005, "colors": 0.0025, "shs": 0.000125, "opacity": 0.05, "rotation": 0.001}
    for k in ("xyz", "colors", "scaling", "opacity", "rotation", "shs"):
        p, q = getattr(gpu.gaussian, k).cpu(), getattr(cpu.gaussian, k)
        d = (p - q).abs()
        frac = float((d > 1e-5 * (1 + q.abs())).float().mean())
        log("%s: max |device - cpu| = %.3e, fraction beyond 1e-5 = %.4f" % (k, float(d.max()), frac))
        assert frac < 0.02, k                          # a handful of elements: Adam's +-lr step on a round-off-sized gradient
        assert float(d.max()) <= 2.5 * cfg_lr[k] * STEPS, k
        a, b = gpu.optimizer.exp_avg[k].cpu(), cpu.optimizer.exp_avg[k]
        rel = float((a - b).norm() / b.norm())
        log("%s: exp_avg rel-L2 = %.3e" % (k, rel))
        assert float(b.norm()) > 0 and rel < 2e-3, k
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "gpu_log_plumbing.log")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        with open(out, "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass


def _loaded_lograst():
    try:
        with open("/proc/self/maps") as f:
            return any("liblograst.so" in line for line in f)
    except OSError:
        return True

"""lograst: the LoG rasterizer hot path (and the callers either side of it) on MI355X.  See README.md / DESIGN.md."""
import sys
import types


def install_compute_radius():
    """Register this repo's ``compute_radius_module`` as the module ``LoG.cuda.compute_radius`` so that
    ``from LoG.cuda.compute_radius import compute_radius_module`` (/root/reference/LoG/model/level_of_gaussian.py:11,
    executed at import time) resolves to the HIP kernel instead of JIT-compiling the reference's CUDA file -- no
    reference file needs editing.  Call before LoG.model is imported."""
    from .compute_radius import compute_radius_module
    shim = types.ModuleType("LoG.cuda.compute_radius")
    shim.compute_radius_module = compute_radius_module
    shim.__doc__ = "log_amd stand-in for LoG/cuda/compute_radius.py (HIP kernel behind lograst_compute_radius)"
    sys.modules["LoG.cuda.compute_radius"] = shim
    return shim


def install_all(fused_step=False):
    """Everything a LoG process needs, in one call: the LoG.cuda.compute_radius module, then every
    drop-in method assigned onto LoG's own classes (needs LoG importable): LoG.get_all, TensorTree.traverse,
    Counter.update_by_output, SparseOptimizer.step.
    fused_step (opt-in, round 6): the backward of LoG.get_all applies SparseOptimizer's update itself, in the kernel that
    computes the raw gradients (log_amd.get_all.set_fused_step: the update then happens at backward time -- the same result
    for LoG's trainer, one backward per step)."""
    install_compute_radius()
    from . import counter, get_all, lod, sparse_optimizer
    get_all.set_fused_step(bool(fused_step))
    return [m.install() for m in (get_all, lod, counter, sparse_optimizer)]

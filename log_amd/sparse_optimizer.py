"""Drop-in for ``SparseOptimizer.step`` (/root/reference/LoG/model/sparse_optimizer.py:163-196 with
``_single_tensor_adam`` :41-78 and the state gather / scatter :198-249) -- SURVEY 8f row N4.

The reference compacts ``index[flag_vis]``, copies it to the host, gathers both Adam moments of every key, runs
~10 elementwise kernels per key and scatters parameters and moments back.  Here all keys are updated in ONE kernel
launch that skips rows with ``flag_vis == False`` itself (include/lograst.h: lograst_sparse_adam); the only host
work is the learning-rate schedule.  ``global_steps`` stays a device buffer (it is part of the state dict) and is
mirrored by a host-side counter so that the step does not synchronise.

Install with ``log_amd.sparse_optimizer.install()`` (= ``SparseOptimizer.step = step``)."""
import math

import torch

from . import rasterizer as _r

BETA1, BETA2, EPS = 0.9, 0.999, 1e-15      # _single_tensor_adam defaults / the eps passed at sparse_optimizer.py:190


def _host_steps(opt):
    n = getattr(opt, "_lograst_steps", None)
    if n is None:
        n = int(opt.global_steps.item())    # once (and again after load_state_dict)
    return n


def _migrate_state(opt, device):
    """The reference parks the Adam moments in host memory above 50 M / 100 M points (LoG/model/splitter.py:198-204) and
    gathers / scatters the visible rows over PCIe every step (sparse_optimizer.py:198-248) -- a 24 GB-card measure.  An
    MI355X holds 288 GB: moments found on another device are moved next to the parameters, once, and stay there
    (100 M points x 59 floats x 2 moments = 47 GB)."""
    dicts = [opt.exp_avg, opt.exp_avg_sq] + ([opt.max_exp_avg_sq] if getattr(opt, "use_amsgrad", False) else [])
    for d in dicts:
        for key in (list(d.keys) if not isinstance(d, dict) else list(d)):
            t = d[key]
            if t.device != device:
                moved = t.to(device)
                if isinstance(d, dict):
                    d[key] = moved
                else:
                    setattr(d, key, moved)       # BufferDict (nn.Module): replaces the registered buffer


def _lr_of(opt, key, steps):
    if key == "xyz":
        return opt.xyz_scheduler_args(steps)
    if key == "scaling":
        return opt.scaling_scheduler_args(steps)
    return opt.lr_dict[key]


def fused_update(fused, pack, n, radii, g_xyz, g_scaling, g_opacity, g_rotation, g_colors):
    """Called from the backward of ``log_amd.get_all`` (``set_fused_step(True)``): the step's Adam update of the rows with
    ``radii > 0``, applied by the kernel that computes their raw gradients.  Uses the scalars ``step`` would use (the step
    number is the NEXT one: ``step`` itself still advances the counters).  -> True when the update was applied; False
    (nothing touched) when this optimizer cannot be fused: the caller then produces the gradients as usual."""
    opt = fused["optimizer"]
    if getattr(opt, "_lograst_fused_pending", False) or getattr(opt, "_lograst_open_packs", 1) != 1:
        # more than one training get_all since the last step() (a batch of several views: their gradients must be SUMMED
        # before Adam sees them), or a second backward through the same pack: everything goes the ordinary way
        return False
    bufs = fused["bufs"]
    try:
        steps = _host_steps(opt) + 1
        bc1, bc2 = 1 - BETA1 ** steps, 1 - BETA2 ** steps
        device = g_xyz.device
        _migrate_state(opt, device)
        entries = {}
        for key in pack["param_keys"]:
            entries[key] = (bufs[key].data, opt.exp_avg[key], opt.exp_avg_sq[key],
                            opt.max_exp_avg_sq[key] if getattr(opt, "use_amsgrad", False) else None, _lr_of(opt, key, steps) / bc1)
    except (KeyError, AttributeError):
        return False
    with torch.no_grad():
        _r._backend.activate_backward_adam(pack["raw"], n, pack["degree"], pack["campos"], g_xyz, g_scaling, g_opacity,
                                           g_rotation, g_colors, fused["index"], radii, entries, BETA1, BETA2,
                                           math.sqrt(bc2), EPS)
    opt._lograst_fused_pending = True
    return True


def step(self, model, index, params, flag_vis):
    """Same signature and effects as SparseOptimizer.step: rows ``index[flag_vis]`` of every ``getattr(model, key)``
    with a gradient, and of its Adam moments, are updated; ``self.xyz_lr`` and ``self.global_steps`` advance."""
    steps = _host_steps(self) + 1            # read (first call only) BEFORE the device-side increment
    self.global_steps += 1
    self._lograst_steps = steps
    self._lograst_open_packs = 0
    if getattr(self, "_lograst_fused_pending", False):
        # the update of this step was applied by the backward (log_amd.get_all, set_fused_step): what is left is the
        # bookkeeping -- the counters above and the learning rate the trainer reads back (level_of_gaussian.py:394)
        self._lograst_fused_pending = False
        if "xyz" in params:
            self.xyz_lr = self.xyz_scheduler_args(steps)
        if all(getattr(p_, "grad", None) is None for p_ in params.values()):
            return
        # (gradients on the parameters all the same: a further backward ran before this step and went the ordinary way --
        # its update is applied below, with this step's scalars)
    bc1 = 1 - BETA1 ** steps
    bc2 = 1 - BETA2 ** steps
    _migrate_state(self, next(iter(params.values())).device)
    entries = []
    for key, param in params.items():
        if param.grad is None:
            continue
        if key == "xyz":
            lr = self.xyz_scheduler_args(steps)
            self.xyz_lr = lr
        elif key == "scaling":
            lr = self.scaling_scheduler_args(steps)
        else:
            lr = self.lr_dict[key]
        entries.append((getattr(model, key).data, param.data, param.grad, self.exp_avg[key], self.exp_avg_sq[key],
                        self.max_exp_avg_sq[key] if self.use_amsgrad else None, lr / bc1))
    if entries:
        with torch.no_grad():
            _r._backend.sparse_adam(index, flag_vis, entries, BETA1, BETA2, math.sqrt(bc2), EPS)


def _load_state_dict(self, state_dict):
    self._lograst_steps = None
    return self._lograst_load_state_dict(state_dict)


def install():
    """Patch the reference class in place (needs LoG importable)."""
    from LoG.model.sparse_optimizer import SparseOptimizer
    if not hasattr(SparseOptimizer, "_lograst_load_state_dict"):
        SparseOptimizer._lograst_load_state_dict = SparseOptimizer.load_state_dict
        SparseOptimizer.load_state_dict = _load_state_dict
    SparseOptimizer.step = step
    return SparseOptimizer

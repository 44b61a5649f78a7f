"""Generates tests/golden/owner_adam.npz by RUNNING the reference's SparseOptimizer.step
(LoG/model/sparse_optimizer.py:163-196) on CPU (build container only) in the situation of the owner-computes step of
log_amd/dist.py: dense per-attribute gradients for all P rows (the sum over a step's views) and a `seen` mask saying
which rows any view touched; the reference is handed index = arange(P), flag_vis = seen.

    python tests/golden/make_golden_owner.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("LOG_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (REF, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from LoG.model.sparse_optimizer import SparseOptimizer     # noqa: E402  reference code, imported not copied

# bucket column name (log_amd/dist.py) -> the reference model's key
KEYS = {"means3D": "xyz", "scales": "scaling", "rotations": "rotation", "opacities": "opacity", "colors": "colors", "shs": "shs"}
SHAPES = {"xyz": (3,), "scaling": (3,), "rotation": (4,), "opacity": (1,), "colors": (3,), "shs": (15, 3)}
LR = {"xyz": 0.00016, "xyz_final": 0.0000016, "colors": 0.0025, "shs": 0.000125, "scaling": 0.005, "opacity": 0.05,
      "rotation": 0.001, "max_steps": 30000}


def main(P=301, n_steps=3, seed=11):
    g = torch.Generator().manual_seed(seed)
    model = types.SimpleNamespace(**{k: torch.randn(P, *s, generator=g) for k, s in SHAPES.items()})
    opt = SparseOptimizer(list(SHAPES), dict(LR), model, device=torch.device("cpu"), xyz_scale=1.0, use_amsgrad=False)
    out = {"P": np.int32(P), "n_steps": np.int32(n_steps)}
    for b, k in KEYS.items():
        out["init_" + b] = getattr(model, k).numpy().copy()
    index = torch.arange(P)
    for it in range(n_steps):
        seen = torch.rand(P, generator=g) < 0.6
        params = {}
        for b, k in KEYS.items():
            p = torch.nn.Parameter(getattr(model, k).clone())
            p.grad = torch.randn(p.shape, generator=g) * (10.0 ** float(torch.randint(-5, 0, (1,), generator=g)))
            params[k] = p
            out[f"s{it}_grad_{b}"] = p.grad.numpy().copy()
        out[f"s{it}_seen"] = seen.numpy()
        opt.step(model, index, params, seen)
        out[f"s{it}_lr_means3D"] = np.float64(opt.xyz_lr)
        out[f"s{it}_lr_scales"] = np.float64(opt.scaling_scheduler_args(opt.global_steps.item()))
    for b, k in KEYS.items():
        out["final_" + b] = getattr(model, k).numpy()
        out["final_exp_avg_" + b] = opt.exp_avg[k].numpy()
        out["final_exp_avg_sq_" + b] = opt.exp_avg_sq[k].numpy()
    np.savez_compressed(os.path.join(HERE, "owner_adam.npz"), **out)
    print("owner_adam", os.path.getsize(os.path.join(HERE, "owner_adam.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()

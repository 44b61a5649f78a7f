"""Generates tests/golden/getall_*.npz by RUNNING the reference's own gather + activation code on CPU (build
container only): the gathers of LoG.get_all (LoG/model/level_of_gaussian.py:262-296, restated in four lines below
because the method needs a whole LoG object; the full method is exercised by tests/test_log_plumbing_cpu.py) and
Activation.activate_root_return (LoG/model/activation.py:27-44, imported), with gradients from torch autograd.

    python tests/golden/make_golden_getall.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("LOG_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

from LoG.model.activation import Activation   # noqa: E402  reference code, imported not copied


def case(seed, P, n_leaf, n_node, K, degree):
    g = torch.Generator().manual_seed(seed)
    model = {"scaling": torch.randn(P, 3, generator=g) - 3.0, "colors": torch.randn(P, 3, generator=g),
             "xyz": torch.rand(P, 3, generator=g) - 0.5, "opacity": torch.randn(P, 1, generator=g) * 2,
             "rotation": torch.randn(P, 4, generator=g)}
    if K:
        model["shs"] = torch.randn(P, K, 3, generator=g) * 0.3
    model["rotation"][:5] *= 1e-3                                  # short quaternions
    perm = torch.randperm(P, generator=g)
    index, index_node = perm[:n_leaf], perm[n_leaf:n_leaf + n_node]
    camera = {"camera_center": torch.tensor([0.3, -2.5, 0.8])}
    # level_of_gaussian.py:267-281 (fix_parent, training)
    params = {k: nn.Parameter(v[index]) for k, v in model.items()}
    full = {k: torch.cat([params[k], v[index_node]]) for k, v in model.items()}
    act = Activation().activate_root_return(full, camera, degree)
    ups = {k: torch.randn(act[k].shape, generator=g) for k in ("xyz", "scaling", "opacity", "rotation", "colors")}
    sum((act[k] * ups[k]).sum() for k in ups).backward()
    out = {"degree": np.int32(degree), "K": np.int32(K), "index": index.numpy(), "index_node": index_node.numpy(),
           "camera_center": camera["camera_center"].numpy()}
    for k, v in model.items():
        out["model_" + k] = v.numpy()
    for k in ups:
        out["act_" + k] = act[k].detach().numpy()
        out["up_" + k] = ups[k].numpy()
    for k, p in params.items():
        out["has_grad_" + k] = np.int32(p.grad is not None)
        if p.grad is not None:
            out["grad_" + k] = p.grad.numpy()
    return out


def main():
    for name, kw in (("deg0", dict(seed=1, P=2500, n_leaf=900, n_node=200, K=3, degree=0)),
                     ("deg1", dict(seed=2, P=2500, n_leaf=1000, n_node=130, K=3, degree=1)),
                     ("deg2", dict(seed=5, P=900, n_leaf=333, n_node=41, K=8, degree=2)),
                     ("deg3", dict(seed=3, P=1500, n_leaf=700, n_node=0, K=15, degree=3)),
                     ("nosh", dict(seed=4, P=800, n_leaf=300, n_node=50, K=0, degree=0))):
        f = os.path.join(HERE, f"getall_{name}.npz")
        np.savez_compressed(f, **case(**kw))
        print(name, os.path.getsize(f) // 1024, "KiB")


if __name__ == "__main__":
    main()

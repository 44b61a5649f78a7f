"""Generates tests/golden/train_*.npz by RUNNING the reference's own bookkeeping code on CPU (build container only):
  * Counter.update_by_output          (LoG/model/counter.py:36-68)            -> counter_*.npz
  * SparseOptimizer.step              (LoG/model/sparse_optimizer.py:163-196) -> adam_*.npz
  * torch.unique(point_id_pixel, ...) (LoG/render/renderer.py:156-159)        -> inside counter_*.npz
The per-view inputs (radii, point_weight, point_id_pixel) come from the oracle's rendering of a small scene; the
gradients are random.  Everything the reference code computes is stored next to its inputs.

    python tests/golden/make_golden_train.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("LOG_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (REF, ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from LoG.model.counter import Counter                      # noqa: E402  reference code, imported not copied
from LoG.model.sparse_optimizer import SparseOptimizer     # noqa: E402
from oracle import oracle                                  # noqa: E402
from log_amd import scenes                                 # noqa: E402

COUNTER_KEYS = ["weights_max", "weights_sum", "grad_sum", "radii_max", "visible_count", "radii_max_max", "area_sum",
                "create_steps"]


def counter_case(seed, P, W, H, n_views):
    import math
    rng = np.random.default_rng(seed)
    sc = scenes.random_scene(P, seed=seed, opacity=None, smax=0.05)
    cams = scenes.orbit_cameras(8, W=W, H=H, focal=0.9 * W)
    counter = Counter(num_points=P)
    out = {"P": np.int32(P), "n_views": np.int32(n_views)}
    views = {"render": [], "visibility_flag": [], "viewspace_points": [], "radii": [], "point_weight": [],
             "point_id": [], "point_count": []}
    for v in range(n_views):
        cam = cams[(3 * v) % 8]
        perm = rng.permutation(P)
        n_leaf, n_node = int(0.5 * P), int(0.1 * P)
        index, index_node = perm[:n_leaf], perm[n_leaf:n_leaf + n_node]
        vis = np.concatenate([index, index_node])
        tfx, tfy = math.tan(cam["FoVx"] * 0.5), math.tan(cam["FoVy"] * 0.5)
        view = oracle.make_view(W, H, tfx, tfy, cam["world_view_transform"], cam["full_proj_transform"], [1, 1, 1])
        f = oracle.forward(view, sc["xyz"][vis], sc["scaling"][vis], sc["rotation"][vis], sc["opacity"][vis],
                           sc["colors"][vis])
        pid_map = torch.from_numpy(f["point_id_pixel"])
        point_id, point_count = torch.unique(pid_map, sorted=True, return_counts=True)      # renderer.py:156
        if point_id[0] == -1:
            point_id, point_count = point_id[1:], point_count[1:]
        grad = (rng.standard_normal((vis.shape[0], 3)) * 1e-3).astype(np.float32)
        vsp = types.SimpleNamespace(grad=torch.from_numpy(grad))
        views["render"].append(None)
        views["visibility_flag"].append({"index": torch.from_numpy(index), "index_node": torch.from_numpy(index_node)})
        views["viewspace_points"].append(vsp)
        views["radii"].append(torch.from_numpy(f["radii"]))
        views["point_weight"].append(torch.from_numpy(f["point_weight"].copy()))
        views["point_id"].append(point_id)
        views["point_count"].append(point_count)
        out.update({f"v{v}_visible_index": vis.astype(np.int64), f"v{v}_grad": grad, f"v{v}_radii": f["radii"],
                    f"v{v}_point_weight": f["point_weight"], f"v{v}_pid_map": f["point_id_pixel"],
                    f"v{v}_point_id": point_id.numpy(), f"v{v}_point_count": point_count.numpy()})
    counter.update_by_output(views, fix_parent=True)
    for v in range(n_views):
        out[f"v{v}_flag_vis"] = views["visibility_flag"][v]["flag_vis"].numpy()
    for k in COUNTER_KEYS:
        out["final_" + k] = getattr(counter, k).numpy()
    return out


LR = {"xyz": 0.00016, "xyz_final": 0.0000016, "colors": 0.0025, "shs": 0.000125, "scaling": 0.005, "opacity": 0.05,
      "rotation": 0.001, "max_steps": 30000}
SHAPES = {"xyz": (3,), "colors": (3,), "scaling": (3,), "opacity": (1,), "rotation": (4,), "shs": (15, 3)}


def adam_case(seed, P, n_steps, amsgrad):
    g = torch.Generator().manual_seed(seed)
    model = types.SimpleNamespace(**{k: torch.randn(P, *s, generator=g) for k, s in SHAPES.items()})
    keys = list(SHAPES)
    opt = SparseOptimizer(keys, dict(LR), model, device=torch.device("cpu"), xyz_scale=1.0, use_amsgrad=amsgrad)
    opt.global_steps += 40            # not the first step: both bias corrections away from their limits
    out = {"P": np.int32(P), "n_steps": np.int32(n_steps), "amsgrad": np.int32(amsgrad),
           "start_global_steps": np.float32(opt.global_steps.item())}
    for k in keys:
        out["init_" + k] = getattr(model, k).numpy().copy()
    for it in range(n_steps):
        m = int(0.4 * P)
        index = torch.randperm(P, generator=g)[:m]
        flag_vis = torch.rand(m, generator=g) < 0.8
        params = {}
        for k in keys:
            p = torch.nn.Parameter(getattr(model, k)[index].clone())
            p.grad = torch.randn(p.shape, generator=g) * (10.0 ** float(torch.randint(-6, 0, (1,), generator=g)))
            params[k] = p
        if it == 1:
            params["rotation"].grad = None       # sparse_optimizer.py:172-173: keys without a gradient are skipped
        out.update({f"s{it}_index": index.numpy(), f"s{it}_flag_vis": flag_vis.numpy()})
        for k in keys:
            out[f"s{it}_param_{k}"] = params[k].data.numpy().copy()
            if params[k].grad is not None:
                out[f"s{it}_grad_{k}"] = params[k].grad.numpy().copy()
        opt.step(model, index, params, flag_vis)
        out[f"s{it}_lr_xyz"] = np.float64(opt.xyz_lr)
    for k in keys:
        out["final_" + k] = getattr(model, k).numpy()
        out["final_exp_avg_" + k] = opt.exp_avg[k].numpy()
        out["final_exp_avg_sq_" + k] = opt.exp_avg_sq[k].numpy()
        if amsgrad:
            out["final_max_exp_avg_sq_" + k] = opt.max_exp_avg_sq[k].numpy()
    out["final_global_steps"] = np.float32(opt.global_steps.item())
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "counter_a.npz"), **counter_case(seed=5, P=6000, W=160, H=120, n_views=3))
    np.savez_compressed(os.path.join(HERE, "adam_a.npz"), **adam_case(seed=6, P=600, n_steps=3, amsgrad=False))
    np.savez_compressed(os.path.join(HERE, "adam_ams.npz"), **adam_case(seed=7, P=300, n_steps=2, amsgrad=True))
    for f in ("counter_a", "adam_a", "adam_ams"):
        print(f, os.path.getsize(os.path.join(HERE, f + ".npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()

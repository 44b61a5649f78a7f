"""N2/N3 measurement: the fused LoG.get_all (gather + activations + SH colours, and their backward) against the same
op sequence in torch on the same GPU (what LoG runs today: level_of_gaussian.py:262-296, activation.py:27-44,
sh_utils.py:31-72), for 1 M selected rows out of a 3 M-row model.
    python tools/bench_get_all.py [rows] [max_degree]  -> one JSON line"""
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from log_amd import get_all, _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 3
K = (D + 1) ** 2 - 1
P, n_node = 3 * N, N // 20
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
bufs = {"scaling": torch.randn(P, 3, device=dev, generator=gen) - 3, "colors": torch.randn(P, 3, device=dev, generator=gen),
        "xyz": torch.rand(P, 3, device=dev, generator=gen) - 0.5, "opacity": torch.randn(P, 1, device=dev, generator=gen),
        "rotation": torch.randn(P, 4, device=dev, generator=gen), "shs": torch.randn(P, K, 3, device=dev, generator=gen) * 0.3}
perm = torch.randperm(P, device=dev, generator=gen)
index, index_node = perm[:N], perm[N:N + n_node]
campos = torch.tensor([0.2, 2.4, -0.7], device=dev)
keys = list(bufs)
gaussian = types.SimpleNamespace(keys=keys, active_sh_degree=D, items=lambda: ((k, bufs[k]) for k in keys),
                                 visibility_flag={"index": index, "index_node": index_node})
model = types.SimpleNamespace(gaussian=gaussian, fix_parent=True, training=True)
camera = {"camera_center": campos}
C0, C1 = 0.28209479177387814, 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def torch_get_all():
    params = {k: torch.nn.Parameter(v[index]) for k, v in bufs.items()}
    full = {k: torch.cat([params[k], v[index_node]]) for k, v in bufs.items()}
    colors = full["colors"] * C0 + 0.5
    d = full["xyz"].detach() - campos[None]
    d = d / torch.norm(d, dim=-1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    sh = full["shs"]
    res = -C1 * y * sh[:, 0] + C1 * z * sh[:, 1] - C1 * x * sh[:, 2]
    if D > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[:, 3] + C2[1] * yz * sh[:, 4] + C2[2] * (2.0 * zz - xx - yy) * sh[:, 5]
               + C2[3] * xz * sh[:, 6] + C2[4] * (xx - yy) * sh[:, 7])
        if D > 2:
            res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 8] + C3[1] * xy * z * sh[:, 9]
                   + C3[2] * y * (4 * zz - xx - yy) * sh[:, 10] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 11]
                   + C3[4] * x * (4 * zz - xx - yy) * sh[:, 12] + C3[5] * z * (xx - yy) * sh[:, 13]
                   + C3[6] * x * (xx - 3 * yy) * sh[:, 14])
    act = {"xyz": full["xyz"], "scaling": torch.exp(full["scaling"]), "opacity": torch.sigmoid(full["opacity"]),
           "rotation": torch.nn.functional.normalize(full["rotation"]), "colors": colors + res}
    return params, act


ups = {"xyz": torch.randn(N + n_node, 3, device=dev, generator=gen), "scaling": torch.randn(N + n_node, 3, device=dev, generator=gen),
       "opacity": torch.randn(N + n_node, 1, device=dev, generator=gen), "rotation": torch.randn(N + n_node, 4, device=dev, generator=gen),
       "colors": torch.randn(N + n_node, 3, device=dev, generator=gen)}


def run_fused():
    ret = get_all.get_all(model, camera, None)
    torch.autograd.backward([ret[k] for k in ups], [ups[k] for k in ups])
    return ret, gaussian.visibility_flag["params"]


def run_torch():
    params, act = torch_get_all()
    torch.autograd.backward([act[k] for k in ups], [ups[k] for k in ups])
    return act, params


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


a, pa = run_fused()
b, pb = run_torch()
err = {k: float((a[k] - b[k]).norm() / b[k].norm()) for k in ups}
gerr = {k: float((pa[k].grad - pb[k].grad).norm() / pb[k].grad.norm()) for k in keys}
_lib.profile_enable(True)
_lib.profile_reset()
ms_fused = timed(run_fused)
prof = _lib.profile_read()
_lib.profile_enable(False)
ms_torch = timed(run_torch)
kern = {k: v[0] / max(v[1], 1) for k, v in prof.items()}
row_floats = 14 + 3 * K
fwd_bytes = (N + n_node) * (8 + row_floats * 4 * 2 + 11 * 4)           # index, gather read, raw write, activated write
bwd_bytes = N * (row_floats * 4 + 11 * 4 + (row_floats - 3) * 4 + 3 * K * 4)
print(json.dumps({
    "bench": "get_all", "rows": N, "node_rows": n_node, "model_rows": P, "degree": D, "floats_per_row": row_floats,
    "rel_l2_outputs": err, "rel_l2_param_grads": gerr, "ms_fused_fwd_bwd": ms_fused, "ms_torch_fwd_bwd_same_gpu": ms_torch,
    "speedup": ms_torch / ms_fused, "ms_gather_activate_kernel": kern.get("gather_activate"),
    "ms_activate_bwd_kernel": kern.get("activate_bwd"),
    "GBs_fwd_kernel": fwd_bytes / max(kern.get("gather_activate", 1e-9), 1e-9) / 1e6,
    "GBs_bwd_kernel": bwd_bytes / max(kern.get("activate_bwd", 1e-9), 1e-9) / 1e6}))

"""N4 measurement at the C2 size (1 M Gaussians, 1080p): per view
  * id histogram (lograst_id_histogram) vs torch.unique(sorted, return_counts) on the same id map,
  * Counter.update_by_output: one kernel vs the reference's sequence of indexing ops restated with torch on the GPU,
  * SparseOptimizer.step (xyz, colors, scaling, opacity, rotation + SH degree 3 = 59 floats per row): one kernel vs
    the reference's gather / _single_tensor_adam / scatter sequence restated with torch on the GPU.
    python tools/bench_train_ops.py [N]  -> one JSON line"""
import json
import math
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from log_amd import counter, sparse_optimizer, scenes, _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W, H = 1920, 1080
dev = torch.device("cuda:0")
sc = scenes.random_scene(N, seed=0)
cam = scenes.orbit_cameras(8, W=W, H=H)[0]
T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
rs = GaussianRasterizationSettings(
    image_height=H, image_width=W, tanfovx=math.tan(cam["FoVx"] * 0.5), tanfovy=math.tan(cam["FoVy"] * 0.5),
    bg=T([1.0, 1.0, 1.0]), scale_modifier=1.0, viewmatrix=T(cam["world_view_transform"]),
    projmatrix=T(cam["full_proj_transform"]), sh_degree=0, campos=T(cam["camera_center"]), prefiltered=False, debug=False)
means2D = torch.zeros(N, 3, device=dev, requires_grad=True)
image, radii, pid, pwp, pw = GaussianRasterizer(raster_settings=rs)(
    means3D=T(sc["xyz"]), means2D=means2D, shs=None, colors_precomp=T(sc["colors"]), opacities=T(sc["opacity"]),
    scales=T(sc["scaling"]), rotations=T(sc["rotation"]), cov3D_precomp=None)
(image * torch.rand_like(image)).sum().backward()


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


# ---- id histogram
def torch_unique():
    i, c = torch.unique(pid, sorted=True, return_counts=True)
    if i[0] == -1:
        i, c = i[1:], c[1:]
    return i, c


ids, counts = counter.unique_ids(pid, N)
ri, rc = torch_unique()
hist_ok = bool(torch.equal(ids, ri) and torch.equal(counts, rc))
ms_hist, ms_unique = timed(lambda: counter.unique_ids(pid, N)), timed(torch_unique)

# ---- counter
P = 3 * N
vis_index = torch.randperm(P, device=dev)[:N]
DT = {"weights_max": torch.float32, "weights_sum": torch.float32, "grad_sum": torch.float32, "radii_max": torch.int16,
      "visible_count": torch.int16, "radii_max_max": torch.int32, "area_sum": torch.int32, "create_steps": torch.int32}
mk = lambda: types.SimpleNamespace(**{k: torch.zeros(P, dtype=d, device=dev) for k, d in DT.items()})
out = {"render": [image], "visibility_flag": [{"index": vis_index}], "viewspace_points": [means2D], "radii": [radii],
       "point_weight": [pw], "point_id": [ids], "point_count": [counts]}


def torch_counter(c):
    """counter.py:36-68 as the reference runs it (one view)."""
    visible_index, grad = vis_index, means2D.grad
    grad_norm = torch.norm(grad[:, :2], dim=-1)
    flag_vis = radii > 0
    index_vis = torch.where(flag_vis)[0]
    point_id, point_count = ids.long(), counts
    c.area_sum[visible_index[point_id]] += point_count
    vvi = visible_index[index_vis]
    c.create_steps[vvi] += 1
    c.visible_count[vvi] += 1
    c.weights_max[vvi] = torch.max(c.weights_max[vvi], pw[index_vis])
    c.weights_sum[vvi] += pw[index_vis]
    c.grad_sum[visible_index[point_id]] += grad_norm[point_id] * point_count
    c.radii_max[vvi] = torch.max(c.radii_max[vvi], radii[index_vis].short())
    c.radii_max_max[visible_index[point_id]] = torch.maximum(point_count.int(), c.radii_max_max[visible_index[point_id]])


c_hip, c_ref = mk(), mk()
counter.update_by_output(c_hip, out)
torch_counter(c_ref)
counter_ok = all(bool(torch.equal(getattr(c_hip, k), getattr(c_ref, k))) if not DT[k].is_floating_point else
                 bool(torch.allclose(getattr(c_hip, k), getattr(c_ref, k), rtol=2e-6, atol=0)) for k in DT)
ms_counter, ms_counter_torch = timed(lambda: counter.update_by_output(c_hip, out)), timed(lambda: torch_counter(c_ref))

# ---- sparse Adam
SH = {"xyz": (3,), "colors": (3,), "scaling": (3,), "opacity": (1,), "rotation": (4,), "shs": (15, 3)}
gen = torch.Generator(device=dev).manual_seed(0)
mkmodel = lambda: types.SimpleNamespace(**{k: torch.randn(P, *s, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) for k, s in SH.items()})
index = vis_index
flag_vis = radii > 0
LR = {"colors": 0.0025, "shs": 0.000125, "opacity": 0.05, "rotation": 0.001}


def mkopt(model):
    z = lambda: {k: torch.zeros_like(getattr(model, k)) for k in SH}
    return types.SimpleNamespace(global_steps=torch.tensor(0., device=dev), lr_dict=dict(LR), exp_avg=z(), exp_avg_sq=z(),
                                 use_amsgrad=False, xyz_lr=None, xyz_scheduler_args=lambda s: 1.6e-4,
                                 scaling_scheduler_args=lambda s: 5e-3)


m_hip, m_ref = mkmodel(), mkmodel()
o_hip, o_ref = mkopt(m_hip), mkopt(m_ref)
params = {}
for k in SH:
    p = torch.nn.Parameter(getattr(m_hip, k)[index].clone())
    p.grad = torch.randn(p.shape, device=dev, generator=gen) * 1e-3
    params[k] = p


def torch_adam(opt, model):
    """sparse_optimizer.py:163-249 as the reference runs it (states on the device)."""
    opt.global_steps += 1
    idx = index[flag_vis]
    idx.cpu()
    step = int(opt.global_steps.item())
    ea = {k: opt.exp_avg[k][idx] for k in SH}
    es = {k: opt.exp_avg_sq[k][idx] for k in SH}
    for k, param in params.items():
        lr = 1.6e-4 if k == "xyz" else (5e-3 if k == "scaling" else LR[k])
        p, g = param.data[flag_vis], param.grad[flag_vis]
        ea[k].mul_(0.9).add_(g, alpha=0.1)
        es[k].mul_(0.999).addcmul_(g, g, value=0.001)
        bc1, bc2 = 1 - 0.9 ** step, 1 - 0.999 ** step
        denom = (es[k].sqrt() / math.sqrt(bc2)).add_(1e-15)
        p.add_(-(lr / bc1) * (ea[k] / denom))
        getattr(model, k).data[idx] = p
    for k in SH:
        opt.exp_avg[k][idx] = ea[k]
        opt.exp_avg_sq[k][idx] = es[k]


sparse_optimizer.step(o_hip, m_hip, index, params, flag_vis)
torch_adam(o_ref, m_ref)
# a last-place difference in the update (lr up to 0.05) against parameters that can be arbitrarily close to 0
adam_ok = all(bool(torch.allclose(getattr(m_hip, k), getattr(m_ref, k), rtol=2e-6, atol=2e-8)) for k in SH)
_lib.profile_enable(True)
_lib.profile_reset()
ms_adam = timed(lambda: sparse_optimizer.step(o_hip, m_hip, index, params, flag_vis))
counter.update_by_output(c_hip, out)
counter.unique_ids(pid, N)
prof = _lib.profile_read()
_lib.profile_enable(False)
ms_adam_torch = timed(lambda: torch_adam(o_ref, m_ref), reps=10)
rows = int(flag_vis.sum())
kern = {k: v[0] / max(v[1], 1) for k, v in prof.items()}
adam_bytes = rows * 59 * 4 * (1 + 1 + 2 + 2 + 1) + rows * 9      # grad, param read; m, v read+write; param write; index + flag
print(json.dumps({
    "bench": "train_ops", "gaussians": N, "pixels": W * H, "distinct_ids": int(ids.numel()), "visible_rows": rows,
    "id_histogram": {"ok": hist_ok, "ms_call": ms_hist, "ms_kernels": kern.get("id_histogram"), "ms_torch_unique": ms_unique,
                     "speedup": ms_unique / ms_hist},
    "counter_update": {"ok": counter_ok, "ms_call": ms_counter, "ms_kernel": kern.get("counter_update"),
                       "ms_torch_same_gpu": ms_counter_torch, "speedup": ms_counter_torch / ms_counter},
    "sparse_adam": {"ok": adam_ok, "floats_per_row": 59, "ms_call": ms_adam, "ms_kernel": kern.get("sparse_adam"),
                    "ms_torch_same_gpu": ms_adam_torch, "speedup": ms_adam_torch / ms_adam,
                    "algorithmic_MB": adam_bytes / 1e6,
                    "GBs_kernel": adam_bytes / max(kern.get("sparse_adam", 1e-9), 1e-9) / 1e6}}))

"""N3 measurement (BASELINE.json configs[2]: LoD tree with level selection on): a 4-ary tree grown from R roots over
L levels, one 1080p camera; times per selection
  * lograst_lod_traverse through the drop-in (log_amd/lod.py),
  * the reference's algorithm as LoG runs it today, restated with torch ops on the same GPU (per level: gather ->
    exp / normalize -> compute_radius kernel -> boolean-mask compaction, `.sum() == 0` host sync; the structure of
    LoG/model/tensor_tree.py:131-185 + level_of_gaussian.py:65-88, with this repo's compute_radius kernel),
  * the CPU oracle (numpy + C, all host cores).
    python tools/bench_lod.py [roots] [levels] [min_px]  -> one JSON line"""
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
sys.path.insert(0, os.path.join(ROOT, "tests"))
from log_amd import lod, scenes, _lib  # noqa: E402
from log_amd.compute_radius import compute_radius_module  # noqa: E402
from log_amd.rasterizer import GaussianRasterizationSettings  # noqa: E402
from lod_util import synth_tree  # noqa: E402

R0 = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 8
MIN_PX = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
W, H = 1920, 1080
dev = torch.device("cuda:0")
s = synth_tree(R0, L, 4, split_prob=0.5, hole_prob=0.02, seed=0, root_scale=0.08)   # ~15 M leaves, 20 M points
P, NN = s["xyz"].shape[0], s["tree"].shape[0]
cam = scenes.orbit_cameras(8, W=W, H=H)[1]
tfx, tfy = math.tan(cam["FoVx"] * 0.5), math.tan(cam["FoVy"] * 0.5)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
tree = types.SimpleNamespace(node_index=t(s["node_index"]), tree=t(s["tree"]), max_level=30, min_resolution_pixel=MIN_PX)
act = types.SimpleNamespace(scaling_activation=torch.exp, rotation_activation=torch.nn.functional.normalize)
model = types.SimpleNamespace(xyz=t(s["xyz"]), scaling=t(s["scaling"]), rotation=t(s["rotation"]), activation=act)
rs = GaussianRasterizationSettings(
    image_height=H, image_width=W, tanfovx=tfx, tanfovy=tfy, bg=torch.zeros(3, device=dev), scale_modifier=1.0,
    viewmatrix=t(np.asarray(cam["world_view_transform"], np.float32)),
    projmatrix=t(np.asarray(cam["full_proj_transform"], np.float32)), sh_degree=0, campos=torch.zeros(3, device=dev),
    prefiltered=False, debug=False)
rast = types.SimpleNamespace(raster_settings=rs)
roots = t(s["root_index"])
fx, fy = W / (2 * tfx), H / (2 * tfy)


def torch_radius(index):
    return compute_radius_module.compute_radius(model.xyz[index], torch.exp(model.scaling[index]),
                                                torch.nn.functional.normalize(model.rotation[index]), rs.projmatrix,
                                                rs.viewmatrix, fx, fy, tfx, tfy)


def torch_traverse(max_depth=1000):
    """The level loop as the reference runs it, on the device."""
    node_index, tr = tree.node_index, tree.tree
    keep = (torch_radius(roots) < MIN_PX) | (node_index[roots] == -1)
    out, index, level = [roots[keep]], roots[~keep], 1
    while True:
        if level > tree.max_level or level > max_depth:
            out.append(index)
            break
        child = tr[node_index[index].long()].flatten().long()
        child = child[child != -1]
        keep = (torch_radius(child) < MIN_PX) | (node_index[child] == -1)
        out.append(child[keep])
        if (~keep).sum() == 0:
            break
        index, level = child[~keep], level + 1
    return torch.cat(out)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


got = lod.traverse(tree, model, roots, rast)
ref = torch_traverse()
same_set = bool(torch.equal(torch.sort(got).values, torch.sort(ref).values))
same_order = bool(torch.equal(got, ref))
_lib.profile_enable(True)
_lib.profile_reset()
ms_hip = timed(lambda: lod.traverse(tree, model, roots, rast), 20)
prof = _lib.profile_read()
_lib.profile_enable(False)
ms_torch = timed(torch_traverse, 10)
from oracle import oracle  # noqa: E402  (CPU baseline only)
t0 = time.perf_counter()
want = oracle.lod_traverse(s["node_index"], s["tree"], s["xyz"], s["scaling"], s["rotation"], s["root_index"],
                           cam["full_proj_transform"], cam["world_view_transform"], fx, fy, tfx, tfy, MIN_PX, 30, 1000)
ms_cpu = (time.perf_counter() - t0) * 1e3
exact = bool(np.array_equal(got.cpu().numpy(), want))
# visited = points whose keep/next decision was evaluated (roots + children of expanded nodes)
depth = s["depth"]
sel = got.cpu().numpy()
expanded = np.zeros(P, bool)
parent_of = np.full(P, -1, np.int64)
ni = s["node_index"]
has = ni >= 0
kids = s["tree"][ni[has]]
par = np.repeat(np.nonzero(has)[0], 4).reshape(-1, 4)
parent_of[kids[kids >= 0]] = par[kids >= 0]
anc = parent_of[sel]
while (anc >= 0).any():
    expanded[anc[anc >= 0]] = True
    anc = np.where(anc >= 0, parent_of[np.maximum(anc, 0)], -1)
visited = int(roots.numel()) + int((s["tree"][ni[expanded]] >= 0).sum())
alg_bytes = visited * (40 + 4 + 4) + sel.shape[0] * 8          # attributes + node_index + tree entry, index out
k_ms, k_n = prof.get("lod_traverse", (0.0, 0))
print(json.dumps({
    "bench": "lod_traverse", "points": P, "nodes": NN, "roots": R0, "levels": int(depth.max()), "min_px": MIN_PX,
    "selected": int(sel.shape[0]), "visited": visited, "exact_vs_oracle": exact, "torch_same_order": same_order,
    "torch_same_set": same_set, "ms_hip_call": ms_hip, "ms_hip_kernels": k_ms / max(k_n, 1), "ms_torch_same_gpu": ms_torch,
    "ms_cpu_oracle": ms_cpu, "cpu_cores": os.cpu_count(), "speedup_vs_torch": ms_torch / ms_hip,
    "algorithmic_MB": alg_bytes / 1e6, "GBs_kernels": alg_bytes / max(k_ms / max(k_n, 1), 1e-9) / 1e6}))

"""A0 (LoG/cuda compute_radius drop-in): streaming bandwidth at LoD-tree scale, next to the reference's own
pure-PyTorch twin restated with numpy on the host (the reference function itself is not on the GPU box).
    python tools/bench_radius.py [P] -> one JSON line"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from log_amd import _lib, scenes  # noqa: E402
from log_amd.compute_radius import compute_radius_module  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
cam = scenes.orbit_cameras(8, W=1920, H=1080, focal=2139.0)[0]
sc = scenes.random_scene(P, seed=0)
t = lambda a: torch.tensor(a, device=dev)
xyz, sca, rot = t(sc["xyz"]), t(sc["scaling"]), t(sc["rotation"])
pm, vm = t(cam["full_proj_transform"]), t(cam["world_view_transform"])
tfx, tfy = math.tan(cam["FoVx"] / 2), math.tan(cam["FoVy"] / 2)
fx, fy = 1920 / (2 * tfx), 1080 / (2 * tfy)
for _ in range(3):
    r = compute_radius_module.compute_radius(xyz, sca, rot, pm, vm, fx, fy, tfx, tfy)
_lib.profile_reset()
_lib.profile_enable(True)
for _ in range(20):
    r = compute_radius_module.compute_radius(xyz, sca, rot, pm, vm, fx, fy, tfx, tfy)
torch.cuda.synchronize()
_lib.profile_enable(False)
ms, cnt = _lib.profile_read()["compute_radius"]
us = 1e3 * ms / cnt
alg = 48.0 * P   # 44 B read + 4 B written per Gaussian
# CPU oracle (C, OpenMP) on the same inputs
from oracle import oracle  # noqa: E402
n_cpu = min(P, 4_000_000)
oracle.lib()
t0 = time.perf_counter()
oracle.compute_radius(sc["xyz"][:n_cpu], sc["scaling"][:n_cpu], sc["rotation"][:n_cpu], cam["full_proj_transform"],
                      cam["world_view_transform"], fx, fy, tfx, tfy)
cpu_s = time.perf_counter() - t0
print(json.dumps({"op": "compute_radius (A0)", "points": P, "avg_launch_us": us, "gaussians_per_s": P / (us * 1e-6),
                  "roofline": {"bound": "hbm", "achieved": alg / (us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": alg / (us * 1e-6) / 1e9 / 8000.0, "algorithmic_bytes_per_launch": alg},
                  "visible_fraction": float((r > 0).float().mean().item()),
                  "cpu_baseline": {"kind": "port", "points": n_cpu, "gaussians_per_s": n_cpu / cpu_s,
                                   "cores": os.cpu_count()}}))

"""Timing of the N1 op (simple_knn.distCUDA2 drop-in) against scipy's cKDTree on the host cores.
    python tools/bench_knn.py [P] -> one JSON line"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_knn._C import distCUDA2  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rng = np.random.default_rng(0)
centers = rng.normal(size=(200, 3)) * 5
pts = (centers[rng.integers(0, 200, P)] + rng.normal(size=(P, 3)) * 0.2 * np.array([1, 1, 0.05])).astype(np.float32)
x = torch.tensor(pts, device="cuda:0")
distCUDA2(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    out = distCUDA2(x)
torch.cuda.synchronize()
gpu_s = (time.perf_counter() - t0) / 3
from scipy.spatial import cKDTree  # noqa: E402
n_cpu = min(P, 500_000)
t0 = time.perf_counter()
tree = cKDTree(pts[:n_cpu].astype(np.float64))
tree.query(pts[:n_cpu].astype(np.float64), k=4, workers=-1)
cpu_s = time.perf_counter() - t0
print(json.dumps({"op": "distCUDA2 (mean sq. dist to 3-NN)", "points": P, "gpu_ms": 1e3 * gpu_s,
                  "gpu_points_per_s": P / gpu_s, "cpu_oracle": "scipy cKDTree build+query k=4, all cores",
                  "cpu_points": n_cpu, "cpu_points_per_s": n_cpu / cpu_s, "cores": os.cpu_count()}))

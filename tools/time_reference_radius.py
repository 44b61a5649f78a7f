"""The reference's OWN pure-PyTorch piece of the path, timed on the host cores of the box it runs on (BASELINE.json north_star:
"alongside the reference's pure-PyTorch/CPU render path timed on the same box's host cores (core count stated)"; SURVEY 8d:
"additionally time the reference's own geometry.compute_radius (geometry.py:132-151) at P = 1 M").  The reference has no CPU
renderer; LoG.model.geometry.compute_radius -- the Python twin of LoG/cuda/compute_radius_kernel.cu, A0 of the scope table
-- is its only CPU-runnable arithmetic on this path.  IMPORTS the reference (LOG_REFERENCE or /root/reference: staged on
the GPU box by tools/run_reference_on_gpu.sh), copies nothing; also runs this framework's A0 kernel on the same inputs when
a GPU is there and reports the agreement.
    python tools/time_reference_radius.py [--points 1000000] [--out gpurun_out/reference_python_radius.json]"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("LOG_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from LoG.model import geometry                    # the reference's code, imported where it lies
    from LoG.dataset.base import prepare_camera
    P = a.points
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    torch.manual_seed(0)                              # SURVEY 8d's C2 recipe (apps/check_gui.py:8-15)
    xyz = torch.rand(P, 3) - 0.5
    scaling = torch.rand(P, 3) * (0.5 * P ** (-1.0 / 3.0))
    rotation = torch.nn.functional.normalize(torch.rand(P, 4))
    th = 0.0
    st, ct = math.sin(th), math.cos(th)
    Rm = np.array([[-st, ct, 0.0], [0.0, 0.0, -1.0], [-ct, -st, 0.0]])
    T = -Rm @ np.array([3.0 * ct, 3.0 * st, 0.0]).reshape(3, 1)
    cam = prepare_camera({"R": Rm, "T": T, "K": np.array([[2139, 0, 960], [0, 2139, 540], [0, 0, 1]], dtype=np.float64),
                          "W": 1920, "H": 1080, "center": (-Rm.T @ T)}, 1, 0.1, 100.0)
    camt = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    # torch's intra-op pool at the box's full core count is not its fastest setting (256 threads in a container: 13.5 s for
    # 1 M points against 0.1 s on 8): sweep the thread count and report the BEST, with the threads it used
    sweep = {}
    with torch.no_grad():
        for nt in sorted({t for t in (4, 8, 16, 32, 64, 128, cores) if t <= cores}):
            torch.set_num_threads(nt)
            r = geometry.compute_radius(xyz, scaling, rotation, camt)      # warm-up (allocator, thread pool)
            times = []
            for _ in range(a.reps):
                t0 = time.perf_counter()
                r = geometry.compute_radius(xyz, scaling, rotation, camt)
                times.append(time.perf_counter() - t0)
            sweep[nt] = min(times)
            if sweep[nt] > 4.0 * min(sweep.values()):
                break                                                          # (far past the optimum: stop burning minutes)
    best_threads = min(sweep, key=sweep.get)
    best = sweep[best_threads]
    res = {"what": "LoG.model.geometry.compute_radius (reference, geometry.py:132-151), CPU PyTorch, best over torch thread counts",
           "points": P, "host_cores": cores, "cores": best_threads, "reps": a.reps,
           "best_s": best, "seconds_by_threads": {str(k): v for k, v in sweep.items()}, "gaussians_per_s": P / best,
           "torch": torch.__version__, "host": os.uname().nodename}
    if torch.cuda.is_available():
        # the same quantity from this framework's A0 kernel (lograst_compute_radius) on the same inputs
        from log_amd import rasterizer as R
        dev = torch.device("cuda:0")
        tfx, tfy = math.tan(cam["FoVx"] * 0.5), math.tan(cam["FoVy"] * 0.5)
        d = lambda x: torch.as_tensor(np.ascontiguousarray(x, np.float32) if isinstance(x, np.ndarray) else x, dtype=torch.float32, device=dev)
        args = (d(xyz), d(scaling), d(rotation), d(cam["full_proj_transform"]), d(cam["world_view_transform"]),
                1920 / (2 * tfx), 1080 / (2 * tfy), tfx, tfy)
        g = R._backend.compute_radius(*args)
        torch.cuda.synchronize()
        best_g = float("inf")
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g = R._backend.compute_radius(*args)
            e1.record()
            e1.synchronize()
            best_g = min(best_g, e0.elapsed_time(e1) * 1e-3)
        g = g.cpu()
        keep = g > 0                                   # (the kernel also applies the 1.3 NDC cull, the Python twin does not)
        rel = float(((g[keep] - r[keep]).abs() / r[keep].clamp_min(1e-6)).max()) if keep.any() else None
        res.update(gpu_kernel_s=best_g, gpu_gaussians_per_s=P / best_g, gpu_over_reference_python=best / best_g,
                   visible=int(keep.sum()), max_rel_diff_on_visible=rel)
    line = json.dumps(res)
    print(line, flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(json.dumps(res, indent=1) + "\n")


if __name__ == "__main__":
    main()

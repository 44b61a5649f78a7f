"""C3-shaped measurement (BASELINE.json configs[2] minus LoG's own LoD tree code): N Gaussians with degree-D spherical
harmonics handed to the drop-in through its native `shs=` input, 1080p, forward+backward including SH evaluation
and dL/dSH.     python tools/bench_sh.py [N] [degree] [views]  -> one JSON line"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from log_amd import rasterizer as R, scenes  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 3
V = int(sys.argv[3]) if len(sys.argv) > 3 else 8
W, H = 1920, 1080
dev = torch.device("cuda:0")
sc = scenes.random_scene(N, seed=0)
T = lambda a, g=False: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev, requires_grad=g)
M = (D + 1) ** 2
rng = np.random.default_rng(2)
leaves = dict(means3D=T(sc["xyz"], True), scales=T(sc["scaling"], True), rotations=T(sc["rotation"], True),
              opacities=T(sc["opacity"], True),
              shs=T((rng.standard_normal((N, M, 3)) * 0.3).astype(np.float32), True))
cams = scenes.orbit_cameras(V, W=W, H=H, focal=2139.0, end_deg=360.0 * (1 - 1.0 / V))
w = torch.rand(3, H, W, device=dev)
rasts = []
for cam in cams:
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam["FoVx"] * 0.5), tanfovy=math.tan(cam["FoVy"] * 0.5),
        bg=T([1.0, 1.0, 1.0]), scale_modifier=1.0, viewmatrix=T(cam["world_view_transform"]),
        projmatrix=T(cam["full_proj_transform"]), sh_degree=D, campos=T(cam["camera_center"]), prefiltered=False,
        debug=False)
    rasts.append(GaussianRasterizer(raster_settings=rs))


sink = {k: torch.zeros_like(v) for k, v in leaves.items()}   # running gradient sums of the step


def one(rast):
    m2 = torch.zeros(N, 3, device=dev, requires_grad=True)
    with R.accumulate_grads_into(sink):
        out = rast(means3D=leaves["means3D"], means2D=m2, shs=leaves["shs"], colors_precomp=None,
                   opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
                   cov3D_precomp=None)
        out[0].backward(gradient=w)


caps = []
for rast in rasts:
    one(rast)
    n, over, mlen, _ = R.last_state_info()
    caps.append((n, mlen))
R.set_instance_capacity(int(max(c[0] for c in caps) * 1.02) + 1024, max_tile_len=int(max(c[1] for c in caps) * 1.02) + 64)
for v in sink.values():
    v.zero_()
for rast in rasts:
    one(rast)
torch.cuda.synchronize()
t0 = time.perf_counter()
for rast in rasts:
    one(rast)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"workload": f"{N} Gaussians, SH degree {D} through shs=, {W}x{H}, {V} orbit views, fwd+bwd (single stream)",
                  "ms_per_view": 1e3 * dt / V, "gaussians_per_s": N * V / dt,
                  "tile_instances_per_view": float(np.mean([c[0] for c in caps]))}))

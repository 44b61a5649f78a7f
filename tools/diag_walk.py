"""Diagnostic (GPU): how far do the four quadrant waves of every tile walk their tile list in the blend kernels?
    python tools/diag_walk.py [gaussians] [opacity|-1]
forward: a wave stops when all of its 64 pixels are saturated, otherwise it walks the whole list; backward: it walks back
from the deepest contributor of its quadrant.  Prints one JSON line with distributions (entries)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G  # noqa: E402
from log_amd import rasterizer as R, scenes  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
OP = float(sys.argv[2]) if len(sys.argv) > 2 else 0.999
W, H = 1920, 1080
dev = torch.device("cuda:0")
cam = scenes.orbit_cameras(8, W=W, H=H)[0]
sc = scenes.random_scene(N, seed=0, opacity=None if OP < 0 else OP)
rs = G.settings(cam, (1, 1, 1), dev)
t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
m, s, r, o, c = t(sc["xyz"]), t(sc["scaling"]), t(sc["rotation"]), t(sc["opacity"]).reshape(-1), t(sc["colors"])
image, radii, pid, pwp, pw, saved = R._backend.forward(rs, R.WODILATE, True, m, s, r, o, c)
torch.cuda.synchronize()
offs = R.tile_offsets_of(saved, W, H).to(torch.int64)
L = (offs[1:] - offs[:-1])
gx, gy = (W + 15) // 16, (H + 15) // 16
nc = torch.zeros(gy * 16, gx * 16, dtype=torch.int64, device=dev)
nc[:H, :W] = saved["n_contrib"]
fT = torch.zeros(gy * 16, gx * 16, device=dev)          # outside the image: "done" from the start
fT[:H, :W] = saved["final_T"]
q_nc = nc.view(gy, 2, 8, gx, 2, 8).amax(dim=(2, 5))      # [gy, 2, gx, 2]: deepest contributor per quadrant
q_open = (fT.view(gy, 2, 8, gx, 2, 8) >= 0.01).any(dim=2).any(dim=-1)   # some pixel of the quadrant (probably) never stopped
Lq = L.view(gy, 1, gx, 1).expand(gy, 2, gx, 2)
fwd_walk = torch.where(q_open, Lq, torch.minimum(q_nc + 64, Lq)).reshape(-1).float()
bwd_walk = q_nc.reshape(-1).float()
pc = lambda x: [float(v) for v in torch.quantile(x, torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=dev))]
act = (Lq.reshape(-1) > 0)
print(json.dumps({
    "gaussians": N, "opacity": OP, "tiles_nonempty": int((L > 0).sum()), "I": int(offs[-1]),
    "list_len_p50_p90_p99_p999_max": pc(L[L > 0].float()),
    "fwd_walk_p50_p90_p99_p999_max": pc(fwd_walk[act]), "fwd_walk_sum": float(fwd_walk.sum()),
    "fwd_open_quadrants": int((q_open.reshape(-1) & act).sum()), "quadrants": int(act.sum()),
    "bwd_walk_p50_p90_p99_p999_max": pc(bwd_walk[act]), "bwd_walk_sum": float(bwd_walk.sum()),
}))

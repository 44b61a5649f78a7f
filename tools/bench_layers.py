"""Sort time on a scene whose per-tile depths cluster in two thin layers (see tests/test_gpu_robustness.py)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
from log_amd import rasterizer as R, scenes, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
rng = np.random.default_rng(9)
cam = scenes.orbit_cameras(8, W=1920, H=1080)[0]
sc = scenes.random_scene(n, seed=9, opacity=None)
layer = np.where(rng.random(n) < 0.5, 0.45, -0.45).astype(np.float32)
sc["xyz"][:, 0] = layer + (rng.standard_normal(n) * 2e-3).astype(np.float32)
dev = torch.device("cuda:0")
rs = G.settings(cam, (0, 0, 0), dev)
t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
m, s, r, o, c = t(sc["xyz"]), t(sc["scaling"]), t(sc["rotation"]), t(sc["opacity"]).reshape(-1), t(sc["colors"])
R._backend.forward(rs, R.WODILATE, True, m, s, r, o, c); torch.cuda.synchronize()
_lib.profile_reset(); _lib.profile_enable(True)
for _ in range(3):
    R._backend.forward(rs, R.WODILATE, True, m, s, r, o, c); torch.cuda.synchronize()
_lib.profile_enable(False)
print(os.environ.get("TAGX", ""), n, R.last_state_info(), {k: round(1e3 * v[0] / v[1]) for k, v in _lib.profile_read().items() if k.startswith("sort")})

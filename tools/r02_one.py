"""One forward at N Gaussians (random scene, view 0), exact mode; prints status.  For kernel printf experiments."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
from log_amd import rasterizer as R, scenes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
dev = torch.device("cuda:0")
cam = scenes.orbit_cameras(8, W=1920, H=1080)[0]
sc = scenes.random_scene(N, seed=0, opacity=0.999)
rs = G.settings(cam, (1, 1, 1), dev)
t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
m, s, r, o, c = t(sc["xyz"]), t(sc["scaling"]), t(sc["rotation"]), t(sc["opacity"]).reshape(-1), t(sc["colors"])
from log_amd import _lib
out = R._backend.forward(rs, R.WODILATE, True, m, s, r, o, c)
torch.cuda.synchronize()
_lib.profile_reset(); _lib.profile_enable(True)
for _ in range(3):
    out = R._backend.forward(rs, R.WODILATE, True, m, s, r, o, c, scratch_floats=7)
    torch.cuda.synchronize()
_lib.profile_enable(False)
print(os.environ.get("TAG", ""), R.last_state_info(), {k: round(1e3 * v[0] / v[1]) for k, v in _lib.profile_read().items()})

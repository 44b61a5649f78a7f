import sys, math, numpy as np, torch, ctypes
sys.path.insert(0, "/root/repo")
from log_amd import _lib, scenes, rasterizer as R
sys.path.insert(0, "/root/repo/tests")
import gpu_util as G
N=1_000_000
sc = scenes.random_scene(N, seed=0); cam = scenes.orbit_cameras(8)[0]
dev = torch.device("cuda:0")
rs = G.settings(cam, (1,1,1), dev)
t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
m, s, r, o, c = t(sc["xyz"]), t(sc["scaling"]), t(sc["rotation"]), t(sc["opacity"]).reshape(-1), t(sc["colors"])
B = R._backend; L = B.require(dev)
view, keep = B.make_view(rs, R.WODILATE, True, dev)
radii = torch.empty(N, dtype=torch.int32, device=dev); geom = torch.empty(L.lograst_geom_bytes(N)//4, device=dev)
state = torch.empty(L.lograst_tile_state_bytes(1920,1080,N)//4, dtype=torch.int32, device=dev)
_lib.profile_reset(); _lib.profile_enable(True)
for _ in range(10):
    _lib.check(L.lograst_forward_project(ctypes.byref(view), N, R._ptr(m), R._ptr(s), R._ptr(r), R._ptr(o), R._ptr(c), R._ptr(radii), R._ptr(geom), R._ptr(state), None, None, R._stream_ptr(dev)))
torch.cuda.synchronize()
print({k: round(v[0] * 1e3 / v[1], 1) for k, v in _lib.profile_read().items()})

import sys, math, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from log_amd import _lib, scenes, rasterizer as R
import gpu_util as G
N=1_000_000
sc = scenes.random_scene(N, seed=0); cam = scenes.orbit_cameras(8)[0]
for fl in (R.WODILATE, R.UPSTREAM):
    for _ in range(2): G.hip_forward(cam, sc, (1,1,1), fl)
    _lib.profile_reset(); _lib.profile_enable(True)
    for _ in range(5): G.hip_forward(cam, sc, (1,1,1), fl)
    torch.cuda.synchronize(); _lib.profile_enable(False)
    print(fl, {k: round(v[0]*1e3/v[1],1) for k,v in _lib.profile_read().items()})

"""Projection kernel on an LoD-shaped selection (big splats first, BFS level order) vs the same Gaussians shuffled.
python tools/micro/proj_lod.py   (env: LOGRAST_TILE_CULL, LOGRAST_BATCH)"""
import ctypes, math, os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from log_amd import _lib, scenes, lod, rasterizer as R
from log_amd.rasterizer import GaussianRasterizationSettings
from lod_util import synth_tree
import gpu_util as G
dev = torch.device("cuda:0")
s = synth_tree(40000, 7, 4, split_prob=0.5, hole_prob=0.02, seed=0, root_scale=0.03)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cam = scenes.orbit_cameras(8)[0]
rs = G.settings(cam, (1, 1, 1), dev)
tree = types.SimpleNamespace(node_index=t(s["node_index"]), tree=t(s["tree"]), max_level=30, min_resolution_pixel=3.0)
act = types.SimpleNamespace(scaling_activation=torch.exp, rotation_activation=torch.nn.functional.normalize)
model = types.SimpleNamespace(xyz=t(s["xyz"]), scaling=t(s["scaling"]), rotation=t(s["rotation"]), activation=act)
idx = lod.traverse(tree, model, t(s["root_index"]), types.SimpleNamespace(raster_settings=rs))
N = int(idx.numel())
B = R._backend; L = B.require(dev)
view, keep = B.make_view(rs, R.WODILATE, True, dev)
gen = torch.Generator(device=dev).manual_seed(0)
for name, order in (("lod_order", idx), ("shuffled", idx[torch.randperm(N, device=dev, generator=gen)])):
    m = model.xyz[order].contiguous(); sc = torch.exp(model.scaling[order]).contiguous()
    r = torch.nn.functional.normalize(model.rotation[order]).contiguous()
    o = torch.full((N,), 0.7, device=dev); c = torch.rand(N, 3, device=dev, generator=gen)
    radii = torch.empty(N, dtype=torch.int32, device=dev); geom = torch.empty(L.lograst_geom_bytes(N) // 4, device=dev)
    state = torch.empty(L.lograst_tile_state_bytes(1920, 1080, N) // 4, dtype=torch.int32, device=dev)
    _lib.profile_reset(); _lib.profile_enable(True)
    for _ in range(5):
        _lib.check(L.lograst_forward_project(ctypes.byref(view), N, R._ptr(m), R._ptr(sc), R._ptr(r), R._ptr(o), R._ptr(c),
                                             R._ptr(radii), R._ptr(geom), R._ptr(state), None, None, R._stream_ptr(dev)))
    torch.cuda.synchronize()
    rad = radii.float()
    nt = ((2 * rad / 16 + 1) ** 2)
    print(name, N, {k: round(v[0] * 1e3 / v[1], 1) for k, v in _lib.profile_read().items()},
          "radius mean/median/max", float(rad.mean()), float(rad.median()), float(rad.max()),
          "frac r>16", float((rad > 16).float().mean()), "approx rect tiles", float(nt.sum()) / 1e6, "M", flush=True)

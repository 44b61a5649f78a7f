"""Rect-size classes of one C3 view (tree-ordered level-of-detail selection): Gaussians and rect-rule tile instances in
rects of 1-4 / 5-16 / 17+ tiles -- what the binning kernels' three paths (LDS ranks / per-lane cursor atomics /
wave-cooperative) each have to place."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import bench_log_step as B
from log_amd import lod, get_all, rasterizer as R

wl = B.Workload(views=4, sh_degree=1)
st = B.State(wl)
rast, camera = wl.rasterizer_for(wl.cams[0])
index_all = lod.traverse(wl.tree, st.gaussian, wl.roots, rast)
index, index_node = B.split_leaf_node(wl, index_all)
st.gaussian.visibility_flag = {"index": index, "index_node": index_node}
act = get_all.get_all(st.model, camera, rast)
with torch.no_grad():
    f = lambda t: t.detach().to(torch.float32).contiguous()
    out = R._backend.forward(rast.raster_settings, R.WODILATE, True, f(act["xyz"]), f(act["scaling"]), f(act["rotation"]),
                             f(act["opacity"]).reshape(-1), f(act["colors"]))
saved = out[-1]
n = act["xyz"].shape[0]
g = saved["geom"][:16 * n].view(-1, 16)
r0, r1 = g[:, 10].view(torch.int32), g[:, 11].view(torch.int32)
w = (r1 & 0xffff) - (r0 & 0xffff); h = (r1 >> 16) - (r0 >> 16)
nt = (w * h).clamp(min=0).cpu().numpy().astype(np.int64)
vis = (saved["radii"] > 0).cpu().numpy()
nt = nt * vis
info = R.last_state_info()
print("Gaussians", n, "visible", int(vis.sum()), "rect-rule instances", int(nt.sum()), "after support cull", info[0])
for lo, hi in ((1, 4), (5, 16), (17, 10**9)):
    m = (nt >= lo) & (nt <= hi)
    print("rects of %d..%s tiles: %9d Gaussians, %10d rect instances (%.1f %%)" % (lo, hi if hi < 10**9 else "", m.sum(), nt[m].sum(), 100.0 * nt[m].sum() / max(nt.sum(), 1)))
# where do the big ones sit in the input order?
big = np.nonzero(nt >= 17)[0]
if len(big):
    print("17+ tile rects: first at row %d, median row %d, last %d of %d; 256-row blocks holding any: %d of %d" % (big[0], np.median(big), big[-1], n, len(np.unique(big // 256)), (n + 255) // 256))

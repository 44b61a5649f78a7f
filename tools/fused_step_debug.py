import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_log_step as B
from log_amd import get_all
wl = B.Workload(roots=3000, levels=4, sh_degree=3, views=3, root_scale=0.05)
packs = [wl.rasterizer_for(c) for c in wl.cams]
states = []
for fused in (False, True):
    st = B.State(wl)
    prev = get_all.set_fused_step(fused)
    try:
        for p in packs[:1]:
            B.view(wl, st, p, True, lambda name: None)
    finally:
        get_all.set_fused_step(prev)
    states.append(st)
a, b = states
for k in wl.keys:
    d = (a.bufs[k] != b.bufs[k])
    print(k, "model diff elems", int(d.sum()), "of", d.numel(), "max abs", float((a.bufs[k] - b.bufs[k]).abs().max()),
          "| m diff", int((a.opt.exp_avg[k] != b.opt.exp_avg[k]).sum()), "v diff", int((a.opt.exp_avg_sq[k] != b.opt.exp_avg_sq[k]).sum()),
          "| moved a", int((a.bufs[k] != wl.bufs[k]).sum()), "moved b", int((b.bufs[k] != wl.bufs[k]).sum()))

"""Where does the HOST time of one view go?  cProfile over a few C2 steps (single stream, sync-free, fused sink).
    python tools/host_profile.py [gaussians]"""
import cProfile, pstats, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from log_amd import rasterizer as R
from log_amd.dist import GradientBucket
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
args = bench.parse.__wrapped__() if hasattr(bench.parse, "__wrapped__") else None
import argparse
a = argparse.Namespace(gaussians=N, views=8, width=1920, height=1080, opacity=0.999)
dev = torch.device("cuda:0")
wl = bench.RasterWorkload(a, N, dev, 0, 1, torch, np)
leaves = {k: v.detach().requires_grad_(True) for k, v in wl.base.items()}
bk = GradientBucket(N, dev, 1); bk.attach(leaves)
out = wl.one_view(wl.rasts[0], leaves); info = R.last_state_info(dev)
R.set_instance_capacity(int(info[0] * 1.1) + 1024, max_tile_len=int(info[2] * 1.1) + 64)
def step():
    with R.accumulate_grads_into(bk.views):
        for rast in wl.rasts:
            wl.one_view(rast, leaves)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue ms/view %.3f   wall ms/view %.3f" % ((t1 - t0) / 80 * 1e3, (t2 - t0) / 80 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])

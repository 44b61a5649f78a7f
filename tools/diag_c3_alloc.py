"""Diagnostic (GPU): where does the host time of a C3 training view go on a fresh box?  Runs the C3 workload of
tools/bench_log_step.py view by view and prints, per stage, the wall time and what torch's caching allocator did
(device allocations / frees / retries, reserved bytes).  python tools/diag_c3_alloc.py [prefill_gb]
prefill_gb: allocate and free that many GB first and call empty_cache() (what bench.py's earlier legs leave behind)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_log_step as B  # noqa: E402

dev = torch.device("cuda:0")
pre = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
if pre > 0:
    blocks = [torch.empty(int(1e9), dtype=torch.uint8, device=dev) for _ in range(int(pre))]
    del blocks
    torch.cuda.empty_cache()
wl = B.Workload(40000, 7, 3, 4, 0.03, dev)
st = B.State(wl)
packs = [wl.rasterizer_for(c) for c in wl.cams]
keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "reserved_bytes.all.current", "allocated_bytes.all.current")
last = {}


def snap():
    s = torch.cuda.memory_stats(dev)
    return {k: s.get(k, 0) for k in keys}


rows = []
cur = {"t": None, "name": None, "m": None}


def clock(name):
    torch.cuda.synchronize()
    now, m = time.perf_counter(), snap()
    if cur["name"] is not None:
        rows.append({"stage": cur["name"], "ms": round(1e3 * (now - cur["t"]), 2),
                     "dalloc": m[keys[0]] - cur["m"][keys[0]], "dfree": m[keys[1]] - cur["m"][keys[1]],
                     "retries": m[keys[2]] - cur["m"][keys[2]], "reserved_gb": round(m[keys[3]] / 1e9, 2),
                     "allocated_gb": round(m[keys[4]] / 1e9, 2)})
    cur["t"], cur["name"], cur["m"] = now, name, m


for rep in range(3):
    for vi, p in enumerate(packs):
        rows.append({"view": vi, "rep": rep})
        B.view(wl, st, p, True, clock)
free, total = torch.cuda.mem_get_info(dev)
print(json.dumps({"prefill_gb": pre, "free_gb": round(free / 1e9, 1), "total_gb": round(total / 1e9, 1), "rows": rows}))

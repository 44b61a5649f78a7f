"""How many rows of the running-sum bucket does a rank's step touch?  (what a row-sparse exchange could skip.)
8 orbit views of the headline scene, gradients added into one row-major bucket: fraction of rows with any non-zero column
after 1, 2, 4, 8 (... 64 with --views 64) views.  python tools/touched_rows.py [--gaussians N] [--opacity X]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=30_000_000)
    ap.add_argument("--opacity", type=float, default=0.999)
    ap.add_argument("--views", type=int, default=8, help="orbit views (64 = the 8 ranks' views together: what the all-gather carries)")
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench as B
    from log_amd import rasterizer as R
    dev = torch.device("cuda:0")
    args = argparse.Namespace(width=1920, height=1080, views=a.views, opacity=a.opacity)
    wl = B.RasterWorkload(args, a.gaussians, dev, 0, 1, torch, np)
    wl.zero_means2d = False
    leaves = {k: v.clone().requires_grad_(True) for k, v in wl.base.items()}
    rows = torch.zeros(wl.N, 16, device=dev)
    out = {}
    with R.accumulate_grads_into({"rows": rows}):
        for i, rast in enumerate(wl.rasts):
            wl.one_view(rast, leaves)
            if i + 1 in (1, 2, 4, 8, 16, 32, 64):
                out[str(i + 1)] = float((rows != 0).any(dim=1).float().mean())
    blocks = {str(b): float((rows != 0).any(dim=1)[: wl.N // b * b].view(-1, b).any(dim=1).float().mean()) for b in (16, 64, 256, 4096)}
    print(json.dumps({"gaussians": wl.N, "opacity": a.opacity, "touched_row_fraction_after_views": out,
                      "touched_block_fraction_after_all_views": blocks}))


if __name__ == "__main__":
    main()

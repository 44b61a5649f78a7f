"""How far into its tile list does the compositing walk go?  One forward per view of bench.py's scenes (raw backend call:
tile offsets, final_T, n_contrib), then per tile: list length L, the last contributor of any of its pixels and whether a
pixel is certainly unsaturated at the end of the list (final_T >= 0.01: alpha is capped at 0.99, so a pixel that stopped
has T < 0.01) -- such a pixel walked all L entries.  Printed per scene: how many long lists (L > WIN positions, the
window of lr_sort_long_kernel) would be finished by their first window alone.  Evidence for DESIGN.md §9 item 2 (lazy
ordering of the lists' tails).
   python tools/walk_depth_probe.py [--gaussians N] [--views V] [--scenes random rand trained]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
WIN = 7680


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=30_000_000)
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scenes", nargs="*", default=["random", "rand", "trained"])
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench as B
    from log_amd import rasterizer as R
    dev = torch.device("cuda:0")
    for scene in a.scenes:
        args = argparse.Namespace(width=a.width, height=a.height, views=a.views, opacity=(-1.0 if scene == "rand" else 0.999),
                                  scene=("trained" if scene == "trained" else "random"))
        wl = B.RasterWorkload(args, a.gaussians, dev, 0, 1, torch, np)
        b = wl.base
        W, H = a.width, a.height
        gx, gy = (W + 15) // 16, (H + 15) // 16
        for vi, rast in enumerate(wl.rasts):
            rs = rast.raster_settings
            image, radii, pid, pwp, pw, saved = R._backend.forward(rs, R.WODILATE, True, b["means3D"], b["scales"], b["rotations"],
                                                                   b["opacities"].reshape(-1), b["colors"])
            torch.cuda.synchronize()
            offs = R.tile_offsets_of(saved, W, H).cpu().numpy().astype(np.int64)
            L = (offs[1:] - offs[:-1])[:gx * gy]
            fT = saved["final_T"].reshape(H, W).cpu().numpy()
            nc = saved["n_contrib"].reshape(H, W).cpu().numpy().astype(np.int64)
            pad = lambda x, fill: np.pad(x, ((0, gy * 16 - H), (0, gx * 16 - W)), constant_values=fill)
            tiles = lambda x: x.reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3).reshape(gy * gx, 256)
            fTt, nct = tiles(pad(fT, 0.0)), tiles(pad(nc, 0))
            open_px = fTt >= 0.01                                   # certainly never stopped: walked the whole list
            walked = np.where(open_px, L[:, None], nct)             # lower bound of the entries each pixel's walk passed
            depth = walked.max(axis=1)                              # how far the tile's slowest pixel went
            long = L > WIN
            need_tail = long & (depth > WIN)
            out = dict(scene=scene, view=vi, tiles=int(gx * gy), instances=int(L.sum()), long_lists=int(long.sum()),
                       keys_in_long_lists=int(L[long].sum()), long_lists_needing_their_tail=int(need_tail.sum()),
                       keys_of_those=int(L[need_tail].sum()),
                       tail_keys_never_needed=int((L[long & ~need_tail] - WIN).sum()),
                       walked_depth_sum=int(depth.sum()), mean_depth_of_finished_long_lists=float(depth[long & ~need_tail].mean()) if (long & ~need_tail).any() else 0.0,
                       open_pixel_tiles=int(open_px.any(axis=1).sum()))
            print(json.dumps(out), flush=True)
        del wl
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

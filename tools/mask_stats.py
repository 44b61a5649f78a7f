"""Design study on the device's own data: how many passes the row-split reverse walk needs per chunk, and how many it would
need if a row's visits were balanced over longer windows.  One training forward (row-split form) of a bench view with a
ZEROED hit-mask buffer; the masks the compositing kernel left are what the reverse walk visits (cut at each row's deepest
contributor).  Per wave and 64-entry chunk: h[row] = visits of the row's 4x4 block; passes of today's loop = max_row
ceil(h / 2).  Simulated: windows of W chunks (passes = max_row ceil(sum_window h / 2)) and windows filled up to 64 wanted
entries (the compact-staging design).
    python tools/mask_stats.py [--gaussians N] [--opacity X] [--scene random|trained] [--view V]"""
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
    ap.add_argument("--scene", default="random")
    ap.add_argument("--view", type=int, default=0)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench as B
    from log_amd import rasterizer as R, tune
    dev = torch.device("cuda:0")
    args = argparse.Namespace(width=a.width, height=a.height, views=8, opacity=a.opacity, scene=a.scene)
    wl = B.RasterWorkload(args, a.gaussians, dev, 0, 1, torch, np)
    W, H = a.width, a.height
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tiles = gx * gy
    R._zero_hit_masks = True
    tune.set_knob("LOGRAST_FWD_ROWS", 1)
    b = wl.base
    with torch.no_grad():
        out = R._backend.forward(wl.rasts[a.view].raster_settings, R.WODILATE, True, b["means3D"], b["scales"], b["rotations"],
                                 b["opacities"].reshape(-1), b["colors"], scratch_floats=16)
    saved = out[-1]
    torch.cuda.synchronize()
    assert saved["hit_masks"] is not None and saved["hit_mask_form"] == 1
    offs = R.tile_offsets_of(saved, W, H).to(torch.int64)
    masks = saved["hit_masks"].view(-1, 4, 4)                       # [slot][wave][block row]
    nc = torch.zeros(gy * 16, gx * 16, dtype=torch.int64, device=dev)
    nc[:H, :W] = saved["n_contrib"]
    # pixel (y, x) -> tile, wave (quadrant), row (4x4 block inside the quadrant)
    nc = nc.view(gy, 2, 2, 4, gx, 2, 2, 4)                           # ty, qy, by, iy, tx, qx, bx, ix
    rmax = nc.amax(dim=(3, 7)).permute(0, 3, 1, 4, 2, 5).reshape(tiles, 4, 4)   # [tile][wave = qy*2+qx][row = by*2+bx]
    maxc = rmax.amax(dim=2)                                          # [tile][wave]
    nch = (maxc + 63) // 64                                          # chunks the wave's reverse walk visits
    K = int(nch.max())
    base = (offs[:-1] // 64) + torch.arange(tiles, device=dev)       # first slot of the tile
    c = torch.arange(K, device=dev)
    slot = (base[:, None] + c[None, :]).clamp_(max=masks.shape[0] - 1)          # [tile][chunk]
    m = masks[slot]                                                  # [tile][chunk][wave][row]
    m = m.permute(0, 2, 1, 3).contiguous()                           # [tile][wave][chunk][row]
    live = c[None, None, :] < nch[:, :, None]                        # chunk visited by this wave
    # cut at the row's deepest contributor: positions 64 c + j < rmax
    lim = (rmax[:, :, None, :] - 64 * c[None, None, :, None]).clamp(0, 64)      # bits [0, lim) stay
    keep = torch.where(lim >= 64, torch.full_like(lim, -1), (torch.ones_like(lim) << lim) - 1)
    m = m & keep
    m = torch.where(live[..., None], m, torch.zeros_like(m))

    def popc(x):
        x = x.clone()
        cnt = torch.zeros_like(x)
        for sh in range(64):
            cnt += (x >> sh) & 1
        return cnt
    h = popc(m)                                                      # [tile][wave][chunk][row]
    union = m[..., 0] | m[..., 1] | m[..., 2] | m[..., 3]
    u = popc(union)
    chunks = int(live.sum())
    res = {"workload": "%d %s Gaussians, opacity %s, view %d" % (a.gaussians, a.scene, a.opacity, a.view),
           "waves_with_work": int((nch > 0).sum()), "wave_chunks": chunks,
           "visits_block_pairs": int(h.sum()), "wanted_entries": int(u.sum()),
           "wanted_entries_per_chunk": float(u.sum()) / chunks, "visits_per_chunk": float(h.sum()) / chunks}
    passes_now = ((h + 1) // 2).amax(dim=3)
    res["passes_now"] = int(passes_now.sum())
    res["passes_now_per_chunk"] = float(passes_now.sum()) / chunks
    res["slot_utilisation_now"] = float(h.sum()) / (8.0 * float(passes_now.sum()))
    # the reverse walk goes from the deepest chunk down: windows are aligned to the deep end
    for Wn in (2, 3, 4, 8):
        tot = 0
        # reverse the chunk axis per wave so that index 0 = deepest visited chunk
        idx = (nch[:, :, None] - 1 - c[None, None, :]).clamp(min=0)
        hr = torch.gather(h, 2, idx[..., None].expand(-1, -1, -1, 4))
        hr = torch.where(live[..., None], hr, torch.zeros_like(hr))
        pad = (-K) % Wn
        if pad:
            hr = torch.cat([hr, torch.zeros(*hr.shape[:2], pad, 4, dtype=hr.dtype, device=dev)], dim=2)
        hw = hr.view(tiles, 4, -1, Wn, 4).sum(dim=3)
        tot = int(((hw + 1) // 2).amax(dim=3).sum())
        res["passes_window_%d" % Wn] = tot
        res["passes_window_%d_per_chunk" % Wn] = tot / chunks
    # windows filled up to 64 wanted entries (greedy from the deep end): CPU loop over the chunk axis, vectorised over waves
    idx = (nch[:, :, None] - 1 - c[None, None, :]).clamp(min=0)
    hr = torch.gather(h, 2, idx[..., None].expand(-1, -1, -1, 4))
    ur = torch.gather(u, 2, idx)
    hr = torch.where(live[..., None], hr, torch.zeros_like(hr))
    ur = torch.where(live, ur, torch.zeros_like(ur))
    for cap in (64, 128):
        acc_u = torch.zeros(tiles, 4, dtype=torch.int64, device=dev)
        acc_h = torch.zeros(tiles, 4, 4, dtype=torch.int64, device=dev)
        passes = torch.zeros(tiles, 4, dtype=torch.int64, device=dev)
        windows = torch.zeros(tiles, 4, dtype=torch.int64, device=dev)
        for k in range(K):
            over = (acc_u + ur[:, :, k]) > cap
            passes += torch.where(over, ((acc_h + 1) // 2).amax(dim=2), torch.zeros_like(passes))
            windows += over.to(torch.int64)
            acc_u = torch.where(over, torch.zeros_like(acc_u), acc_u)
            acc_h = torch.where(over[..., None], torch.zeros_like(acc_h), acc_h)
            acc_u += ur[:, :, k]
            acc_h += hr[:, :, k]
        passes += ((acc_h + 1) // 2).amax(dim=2)
        windows += (acc_u > 0).to(torch.int64)
        res["passes_compact_%d" % cap] = int(passes.sum())
        res["passes_compact_%d_per_chunk" % cap] = int(passes.sum()) / chunks
        res["windows_compact_%d" % cap] = int(windows.sum())
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()

"""Debug aid: a 30 M view with lazily ordered lists, finished (lograst_finish_lists), against the same view with
LOGRAST_LAZY_SORT=0 -- which tiles / positions differ."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench as B
from log_amd import rasterizer as R, tune

def main():
    dev = torch.device("cuda:0")
    args = argparse.Namespace(width=1920, height=1080, views=1, opacity=0.999, scene="random")
    wl = B.RasterWorkload(args, int(os.environ.get("N", 30_000_000)), dev, 0, 1, torch, np)
    b = wl.base; rs = wl.rasts[0].raster_settings; W, H = 1920, 1080
    def fwd():
        prev = R.keep_keys(True)
        try:
            out = R._backend.forward(rs, R.WODILATE, True, b["means3D"], b["scales"], b["rotations"], b["opacities"].reshape(-1), b["colors"])
        finally:
            R.keep_keys(prev)
        torch.cuda.synchronize()
        return out[-1]
    tune.set_knob("LOGRAST_LAZY_SORT", 0)
    s0 = fwd()
    offs = R.tile_offsets_of(s0, W, H).cpu().numpy().astype(np.int64); I = int(offs[-1])
    full = s0["plist"][:I].cpu().numpy()
    tune.set_knob("LOGRAST_LAZY_SORT", 1)
    s1 = fwd()
    offs1 = R.tile_offsets_of(s1, W, H).cpu().numpy().astype(np.int64)
    assert (offs == offs1).all()
    tiles = len(offs) - 1
    st = s1["state"]
    ordered = R.ordered_lengths_of(s1, W, H).cpu().numpy().astype(np.int64)
    lens = np.diff(offs)
    before = s1["plist"][:I].cpu().numpy()
    R.finish_lists(s1, W, H); torch.cuda.synchronize()
    after = s1["plist"][:I].cpu().numpy()
    ordered2 = R.ordered_lengths_of(s1, W, H).cpu().numpy().astype(np.int64)
    print("tiles", tiles, "long", int((lens > 4096).sum()), "lazy", int((ordered < lens).sum()), "after finish incomplete", int((ordered2 < lens).sum()))
    bad_tiles = []
    for t in np.nonzero(lens > 4096)[0]:
        bq, L = int(offs[t]), int(lens[t])
        d = np.nonzero(after[bq:bq + L] != full[bq:bq + L])[0]
        if len(d):
            bad_tiles.append((int(t), L, int(ordered[t]), int(d[0]), int(d[-1]), len(d)))
        dp = np.nonzero(before[bq:bq + int(ordered[t])] != full[bq:bq + int(ordered[t])])[0]
        if len(dp):
            print("PREFIX differs tile", t, L, int(ordered[t]), int(dp[0]), len(dp))
    print("bad tiles", len(bad_tiles))
    for x in bad_tiles[:20]:
        print(" tile %d L %d ordered %d first_diff %d last_diff %d ndiff %d" % x)

main()

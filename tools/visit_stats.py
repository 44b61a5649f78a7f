"""Design study (CPU, numpy): how many wave visits / lane slots the blend kernels spend on the C2 workload under
different pixel-group granularities.  Walks a sample of tiles of one view with the oracle's tile lists and counts
  * (entry, 8x8 quadrant) visits passing the support cull  [current kernels: 64 lane slots each]
  * (entry, 4x4 block) visits passing the same cull       [row-split design: 16 lane slots each]
  * contributing lanes.
python tools/visit_stats.py [n_tiles]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from log_amd import scenes  # noqa: E402
from oracle import oracle  # noqa: E402


def support_min(mx, my, A, B, C, x0, x1, y0, y1):
    """min over the box of 0.5 d^T Q d (numpy, vectorised over entries)."""
    dx0, dx1 = (x0 - 0.01) - mx, (x1 + 0.01) - mx
    dy0, dy1 = (y0 - 0.01) - my, (y1 + 0.01) - my
    inside = (dx0 <= 0) & (dx1 >= 0) & (dy0 <= 0) & (dy1 >= 0)
    best = np.full_like(mx, 3e38)
    for dx in (dx0, dx1):
        dy = np.minimum(dy1, np.maximum(dy0, -B * dx / C))
        best = np.minimum(best, 0.5 * (A * dx * dx + C * dy * dy) + B * dx * dy)
    for dy in (dy0, dy1):
        dx = np.minimum(dx1, np.maximum(dx0, -B * dy / A))
        best = np.minimum(best, 0.5 * (A * dx * dx + C * dy * dy) + B * dx * dy)
    return np.where(inside, 0.0, best)


def main():
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    N, W, H = 1_000_000, 1920, 1080
    sc = scenes.random_scene(N, seed=0)
    cam = scenes.orbit_cameras(8, W=W, H=H)[0]
    tfx, tfy = math.tan(cam["FoVx"] / 2), math.tan(cam["FoVy"] / 2)
    v = oracle.make_view(W, H, tfx, tfy, cam["world_view_transform"], cam["full_proj_transform"], [1, 1, 1])
    f = oracle.forward(v, sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["colors"])
    rec, off, pl, ncon = f["rec"], f["tile_offsets"], f["point_list"], f["n_contrib"]
    gx, gy = oracle.grid(v)
    lens = np.diff(off.astype(np.int64))
    rng = np.random.default_rng(1)
    # sample tiles proportionally to list length (that is where the time goes)
    tiles = rng.choice(len(lens), size=nt, p=lens / lens.sum())
    tot = dict(entries=0, q_visits=0, q_visits_live=0, b_visits=0, b_visits_live=0, lanes=0, q_hit=0, b_hit=0,
               b_rowmax=0, t_visits=0, t_visits_live=0, t_hit=0)
    for t in tiles:
        ids = pl[off[t]:off[t + 1]]
        r = rec[ids]
        mx, my, A, B, C, op = r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4], r[:, 5]
        tau = 1.01 * np.log(255.0 * op) + 0.01
        tx, ty = t % gx, t // gx
        px = tx * 16 + np.arange(16)[None, :].repeat(16, 0)
        py = ty * 16 + np.arange(16)[:, None].repeat(16, 1)
        dxp = mx[:, None, None] - px[None]
        dyp = my[:, None, None] - py[None]
        power = -0.5 * (A[:, None, None] * dxp * dxp + C[:, None, None] * dyp * dyp) - B[:, None, None] * dxp * dyp
        alpha = np.minimum(0.99, op[:, None, None] * np.exp(power))
        ok = (power <= 0) & (alpha >= 1 / 255.0)
        # transmittance walk
        T = np.ones((16, 16))
        done = np.zeros((16, 16), bool)
        acc = np.zeros_like(ok)
        for e in range(len(ids)):
            o = ok[e] & ~done
            test = T * (1 - alpha[e])
            stop = o & (test < 1e-4)
            a = o & ~stop
            T = np.where(a, test, T)
            done |= stop
            acc[e] = a
        inside = (px < W) & (py < H)
        lastc = ncon[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
        last = np.zeros((16, 16), np.int64)
        last[:lastc.shape[0], :lastc.shape[1]] = lastc
        tot["entries"] += len(ids)
        tot["lanes"] += int(acc.sum())
        e_idx = np.arange(len(ids))
        for g, key in ((16, "t"), (8, "q"), (4, "b")):
            per_group_live = []
            for by in range(0, 16, g):
                for bx in range(0, 16, g):
                    x0, x1 = tx * 16 + bx, tx * 16 + bx + g - 1
                    y0, y1 = ty * 16 + by, ty * 16 + by + g - 1
                    hit = support_min(mx, my, A, B, C, x0, x1, y0, y1) <= tau
                    live = hit & (e_idx < last[by:by + g, bx:bx + g].max())   # forward walk stops at the group's last contributor
                    tot[key + "_visits"] += int(hit.sum())
                    tot[key + "_visits_live"] += int(live.sum())
                    tot[key + "_hit"] += int((acc[:, by:by + g, bx:bx + g].any(axis=(1, 2))).sum())
                    per_group_live.append(int(live.sum()))
            if g == 4:
                # row-split: a wave = 8x8 quadrant = four 4x4 blocks walking their own lists; wave time = max over its rows
                pg = np.array(per_group_live).reshape(4, 4)
                for qy in range(2):
                    for qx in range(2):
                        tot["b_rowmax"] += int(pg[2 * qy:2 * qy + 2, 2 * qx:2 * qx + 2].max())
    E = tot["entries"]
    print(f"tiles sampled {nt}, entries {E}")
    print(f"16x16 tile instances passing the support test {tot['t_visits'] / E:.3f}   contributing {tot['t_hit'] / E:.3f}")
    print(f"8x8 visits/entry (cull only)     {tot['q_visits'] / E:.3f}   live (before group's last contributor) {tot['q_visits_live'] / E:.3f}"
          f"   contributing {tot['q_hit'] / E:.3f}")
    print(f"4x4 visits/entry (cull only)     {tot['b_visits'] / E:.3f}   live {tot['b_visits_live'] / E:.3f}   contributing {tot['b_hit'] / E:.3f}")
    print(f"contributing lanes/entry {tot['lanes'] / E:.2f}")
    print(f"lane slots/entry: current 64*{tot['q_visits_live'] / E:.3f} = {64 * tot['q_visits_live'] / E:.1f};  "
          f"row-split 16*{tot['b_visits_live'] / E:.3f} = {16 * tot['b_visits_live'] / E:.1f};  "
          f"row-split wave iterations/entry (max over the 4 rows) {tot['b_rowmax'] / E:.3f} vs current {tot['q_visits_live'] / E:.3f}")


if __name__ == "__main__":
    main()

"""Randomised parity sweep of the HIP path against the CPU oracle (run ON a GPU box, from the repo root):

    python tools/fuzz_parity.py --cases 200 --seed 1 [--out gpurun_out/fuzz.jsonl]

Each case draws a scene (size, scale / opacity statistics, degenerate rows, duplicates), a camera (resolution not a multiple
of 16, focal, distance -- including cameras inside the cloud), a flavour (the fork's `wodilate` 5-tuple / the upstream
2-tuple), a path (plain backend call / the training forward that prepares the accumulator rows and hit masks), the
compositing form of the forward and of the reverse walk (package's choice / rows / quadrant) and a few launch-shape knobs,
then checks what tests/test_gpu_parity.py checks on its fixed cases:
  * forward: records, tile lists, image, final_T, n_contrib and the fork's maps BIT FOR BIT the oracle's;
  * reverse walk (dL/dmean2D, dL/dconic, dL/dopacity, dL/dcolour): rel-L2 <= 1e-4 against the oracle;
  * end to end (dL/dmeans3D, dL/dscales, dL/drotations): the float64-anchored criterion of tests/gpu_util.py.
A failing case is printed with its seed (`--only SEED` replays it) and the run exits 1.  The oracle is the checker here,
as in tests/ -- this is test infrastructure, not product code."""
import argparse
import json
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GRAD_TOL = 1e-4
MAX_INSTANCES = 6_000_000


def draw_case(seed):
    """-> (description dict, cam, scene dict of numpy arrays, options dict)."""
    from log_amd import scenes
    rng = np.random.default_rng(seed)
    n = int(np.exp(rng.uniform(np.log(1), np.log(400000))))
    shape = rng.choice(["odd", "odd", "odd", "aligned", "tiny", "wide", "hd"])
    if shape == "odd":
        W, H = int(rng.integers(17, 700)), int(rng.integers(17, 500))
    elif shape == "aligned":
        W, H = 16 * int(rng.integers(1, 40)), 16 * int(rng.integers(1, 30))
    elif shape == "tiny":
        W, H = int(rng.integers(1, 17)), int(rng.integers(1, 17))
    elif shape == "wide":
        W, H = int(rng.integers(600, 2000)), int(rng.integers(1, 40))
    else:
        W, H = 1920, 1080
        n = min(n, 60000)
    focal = float(W * np.exp(rng.uniform(np.log(0.4), np.log(3.0))))
    radius = float(rng.choice([0.2, 0.8, 1.5, 2.5, 4.0]))
    nv = int(rng.integers(1, 9))
    cam = scenes.orbit_cameras(nv, W=W, H=H, focal=focal, radius=radius)[int(rng.integers(0, nv))]
    kind = rng.choice(["random", "random", "trained", "fine", "coarse"])
    if kind == "coarse":
        n = min(n, 3000)                            # (every splat covers much of the image: the oracle's time is the limit)
    if kind == "trained":
        sc = scenes.trained_like_scene(n, seed=seed % 1000)
    else:
        smax = {"random": float(np.exp(rng.uniform(np.log(0.003), np.log(0.3)))), "fine": 0.002, "coarse": 0.6}[kind]
        sc = scenes.random_scene(n, seed=seed % 1000, smax=smax)
    sc = {k: np.array(v, np.float32, copy=True) for k, v in sc.items()}
    sc["opacity"] = sc["opacity"].reshape(-1)
    omode = rng.choice(["keep", "opaque", "half", "uniform", "fog", "mixed"])
    if omode == "opaque":
        sc["opacity"][:] = 0.999
    elif omode == "half":
        sc["opacity"][:] = 0.5
    elif omode == "uniform":
        sc["opacity"] = rng.random(n, dtype=np.float32)
    elif omode == "fog":
        sc["opacity"] = (0.004 + 0.05 * rng.random(n, dtype=np.float32)).astype(np.float32)
    elif omode == "mixed":
        sc["opacity"] = np.where(rng.random(n) < 0.5, 0.999, 0.02 * rng.random(n)).astype(np.float32)
    degenerate = []
    if n >= 8 and rng.random() < 0.5:
        k = max(1, n // 50)
        pick = lambda: rng.integers(0, n, k)
        for what in rng.choice(["dup", "zero_scale", "zero_opacity", "needle", "giant", "unnormalised"], 2, replace=False):
            ids = pick()
            if what == "dup":                       # equal depths: the (depth, id) order decides
                src = pick()
                for f in ("xyz", "scaling", "rotation"):
                    sc[f][ids] = sc[f][src]
            elif what == "zero_scale":
                sc["scaling"][ids] = 0.0
            elif what == "zero_opacity":
                sc["opacity"][ids] = 0.0
            elif what == "needle":
                sc["scaling"][ids, 1:] *= 1e-3
            elif what == "giant":
                sc["scaling"][ids[: max(1, k // 8)]] = 3.0
            elif what == "unnormalised":            # the reference does not normalise the quaternion in the kernel
                sc["rotation"][ids] *= rng.uniform(0.3, 3.0, (len(ids), 1)).astype(np.float32)
            degenerate.append(str(what))
    opt = dict(flavour=str(rng.choice(["wodilate", "upstream"])), training=bool(rng.random() < 0.6),
               fwd_form=rng.choice([None, "rows", "quadrant"]), bwd_form=rng.choice([None, "rows", "quadrant"]),
               hit_masks=bool(rng.random() < 0.8), use_filter=bool(rng.random() < 0.9),
               scale_modifier=float(rng.choice([1.0, 1.0, 0.5, 2.3])),
               bg=tuple(float(x) for x in rng.random(3).round(2)),
               knobs={})
    for knob, values in (("LOGRAST_LAZY_SORT", (0, 1)), ("LOGRAST_MID_RANK", (0, 1)), ("LOGRAST_MID_COOP", (0, 1, 16)),
                         ("LOGRAST_FILL_STAGED", (0, 1, 2, 3)), ("LOGRAST_BATCH_SLOTS", (64, 256, 1024)),
                         ("LOGRAST_DEFER_TILES", (4, 16, 100)), ("LOGRAST_PBWD_LIST", (0, 1, 2)),
                         ("LOGRAST_HUGE_CHUNK", (256, 2048)), ("LOGRAST_PROJECT_BLOCKS", (64, 4096))):
        if rng.random() < 0.25:
            opt["knobs"][knob] = int(rng.choice(values))
    if opt["fwd_form"] is not None:
        opt["fwd_form"] = str(opt["fwd_form"])
    if opt["bwd_form"] is not None:
        opt["bwd_form"] = str(opt["bwd_form"])
    desc = dict(seed=int(seed), n=n, W=W, H=H, focal=round(focal, 1), radius=radius, scene=str(kind), opacity=str(omode),
                degenerate=degenerate, **{k: v for k, v in opt.items()})
    return desc, cam, sc, opt


def run_case(oracle, seed):
    import gpu_util as G
    from log_amd import rasterizer as R, tune
    from util import rel_l2
    desc, cam, sc, opt = draw_case(seed)
    fl = {"wodilate": R.WODILATE, "upstream": R.UPSTREAM}[opt["flavour"]]
    prev = {k: tune.get_knob(k) for k in opt["knobs"]}
    for k, v in opt["knobs"].items():
        tune.set_knob(k, v)
    try:
        hf = G.hip_forward(cam, sc, opt["bg"], flavour=fl, use_filter=opt["use_filter"], scale_modifier=opt["scale_modifier"],
                           scratch_floats=16 if opt["training"] else 0, fwd_form=opt["fwd_form"],
                           hit_masks=opt["hit_masks"] if opt["training"] else None)
        if hf["I"] > MAX_INSTANCES:
            return dict(desc, I=int(hf["I"]), skipped="more tile instances than the CPU oracle walks in seconds")
        v, of = G.oracle_forward(oracle, cam, sc, opt["bg"], flavour=fl, use_filter=opt["use_filter"],
                                 scale_modifier=opt["scale_modifier"])
        st = G.compare_forward(hf, of)
        bad = {k: x for k, x in st.items() if k not in ("I_hip", "I_oracle", "culled_instances", "lazy_lists") and x != 0}
        assert not bad, ("forward", bad)
        res = dict(desc, I=int(hf["I"]), visible=int((of["radii"] > 0).sum()), fwd_form=hf["fwd_form"],
                   lazy_lists=st["lazy_lists"], culled_instances=st.get("culled_instances", 0))
        dL = np.random.default_rng(seed + 7).standard_normal(of["image"].shape).astype(np.float32)
        hg = G.hip_backward(hf, dL, bwd_form=opt["bwd_form"])
        og = oracle.backward(v, of, dL)
        res.update(bwd_form=hg["bwd_form"], bwd_masks=hg["bwd_masks"])
        finite = all(np.isfinite(og[k]).all() for k in og if k in hg)
        res["oracle_finite"] = bool(finite)
        assert (hg["means2D"][:, 2] == 0).all()
        if not finite:
            for k in ("means3D", "scales", "rotations"):     # non-finite rows must be non-finite in the same places
                assert (np.isfinite(hg[k]) == np.isfinite(og[k])).all(), ("non-finite pattern", k)
            return res
        a = G.gradient_anchor_stats(hg, og, oracle.backward_f64(v, of, dL)) if res["visible"] > 0 else None
        for k in ("means2D", "conic", "opacities", "colors"):
            e = rel_l2(hg[k], og[k])
            res["rel_" + k] = e
            # within the tolerance of the oracle -- or, where ONE Gaussian sums a million signed terms (a splat over the
            # whole image: the fp32 sums of the oracle's pixel order and of the device's tree differ by more than that),
            # no further from the float64 twin than the oracle is (the criterion of tests/gpu_util.py)
            assert e < GRAD_TOL or (a is not None and a[k]["rel_l2_hip"] <= max(GRAD_TOL, 1.25 * a[k]["rel_l2_oracle"])), \
                ("reverse walk", k, e, a[k] if a else None)
        if a is not None:
            for k in ("means3D", "scales", "rotations"):
                s_ = a[k]
                res["rel_all_" + k] = s_["rel_l2_all_hip_vs_oracle"]
                res["excluded_" + k] = s_["excluded_fraction"]       # reported, not bounded: the draws plant degenerate rows
                assert s_["rel_l2_well_hip"] <= max(GRAD_TOL, 1.25 * s_["rel_l2_well_oracle"]), (k, s_)
                assert s_["row_bound_violations"] == 0, (k, s_)
                assert s_["zero_rows_nonzero"] == 0, (k, s_)
                if s_["rows"] > 64:                                  # (in L2 over a handful of rows one row IS the norm)
                    assert s_["err_l2_excess_units"] <= G.L2_FLOOR, (k, s_)
        return res
    finally:
        for k, v in prev.items():
            tune.set_knob(k, v)


def run_props(seed):
    """Size-independent properties on a drawn case, through the packages' autograd drop-in (no oracle):
    (1) the image split into `world` bands of tile rows (log_amd.rasterizer.tile_rows) renders the whole image bit for bit,
        radii / point_weight are the maxima over the bands, and the bands' reverse-walk gradients sum to the whole view's;
    (2) two or three views accumulated by the backward kernels into a gradient bucket (row-major rows / planar attribute
        blocks: accumulate_grads_into) equal autograd's own view-by-view accumulation."""
    import torch
    import gpu_util as G
    from diff_gaussian_rasterization_wodilate import GaussianRasterizer as RastFork
    from diff_gaussian_rasterization import GaussianRasterizer as RastUp
    from log_amd import dist as D, rasterizer as R, scenes
    desc, cam, sc, opt = draw_case(seed)
    rng = np.random.default_rng(seed + 99)
    dev = torch.device("cuda:0")
    n, W, H = desc["n"], desc["W"], desc["H"]
    fork = opt["flavour"] == "wodilate"
    Rast = RastFork if fork else RastUp
    T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
    names = dict(means3D="xyz", scales="scaling", rotations="rotation", opacities="opacity", colors="colors")
    w = torch.tensor(rng.standard_normal((3, H, W)).astype(np.float32), device=dev)
    res = dict(desc)

    def render(camera, rows=None, sink=None, leaves=None):
        leaves = leaves or {k: T(sc[v]).requires_grad_(True) for k, v in names.items()}
        m2 = torch.zeros(n, 3, device=dev, requires_grad=True)
        rast = Rast(raster_settings=G.settings(camera, opt["bg"], dev, opt["scale_modifier"]))
        kw = dict(means3D=leaves["means3D"], means2D=m2, shs=None, colors_precomp=leaves["colors"], opacities=leaves["opacities"],
                  scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
        with (R.tile_rows(*rows) if rows else R.tile_rows(0, 0)):
            if sink is not None:
                with R.accumulate_grads_into(sink):
                    out = rast(**kw)
                    (out[0] * w).sum().backward()
            else:
                out = rast(**kw)
                img = out[0] if rows is None else out[0][:, rows[0] * 16:rows[1] * 16]
                ww = w if rows is None else w[:, rows[0] * 16:rows[1] * 16]
                (img * ww).sum().backward()
        return out, leaves, m2

    # ---- (1) bands ----
    full, lv, m2 = render(cam)
    torch.cuda.synchronize()
    if R.last_state_info()[0] > 40_000_000:
        return dict(res, skipped="more instances than worth rendering several times")
    finite = bool(torch.isfinite(lv["means3D"].grad).all() and torch.isfinite(lv["scales"].grad).all())
    world = int(rng.choice([2, 3, 5, 8]))
    image = torch.empty_like(full[0])
    radii = torch.zeros_like(full[1])
    gs = {k: torch.zeros_like(v.grad) for k, v in lv.items()}
    gm2 = torch.zeros_like(m2.grad)
    pw = torch.zeros_like(full[4]) if fork else None
    for r in range(world):
        rows = D.band_rows(r, world, H)
        b, e = D.band_pixels(r, world, H)
        if rows[1] <= rows[0]:
            continue
        out, l2, mm = render(cam, rows=rows)
        image[:, b:e] = out[0][:, b:e]
        radii = torch.maximum(radii, out[1])
        if fork:
            pw = torch.maximum(pw, out[4])
        for k in gs:
            gs[k] += l2[k].grad
        gm2 += mm.grad
    covered = D.band_pixels(world - 1, world, H)[1] == H and all(D.band_rows(r, world, H)[1] > D.band_rows(r, world, H)[0] for r in range(world))
    if covered:
        assert torch.equal(image, full[0]), "bands: image"
        assert torch.equal(radii, full[1]), "bands: radii"
        assert not fork or torch.equal(pw, full[4]), "bands: point_weight"
        rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
        res["band_rel"] = {k: rel(gs[k], lv[k].grad) for k in ("opacities", "colors")}
        res["band_rel"]["means2D"] = rel(gm2, m2.grad)
        # (a splat over the whole image sums ~1e6 signed terms per Gaussian: the bands' partial sums and the whole view's
        # differ by fp32 summation order -- measured 1.1e-4 on one such case; 3e-4 there, 1e-4 everywhere else)
        tol = 3e-4 if ("giant" in desc["degenerate"] or desc["scene"] == "coarse") else GRAD_TOL
        for k, v in res["band_rel"].items():
            assert v < tol or not finite, ("bands", k, v, world)
    res["bands"] = world if covered else 0
    # ---- (2) accumulation into a bucket ----
    nv = int(rng.integers(2, 4))
    cams = [cam] + [scenes.orbit_cameras(5, W=W, H=H, focal=desc["focal"], radius=desc["radius"])[int(rng.integers(0, 5))] for _ in range(nv - 1)]
    prev = R.set_inplace_leaf_grads(False)
    try:
        leaves = {k: T(sc[v]).requires_grad_(True) for k, v in names.items()}
        for c in cams:
            render(c, leaves=leaves)
        torch.cuda.synchronize()
        want = {k: v.grad.clone() for k, v in leaves.items()}
    finally:
        R.set_inplace_leaf_grads(prev)
    for row_major in (True, False):
        bucket = D.GradientBucket(n, dev, 1, row_major=row_major)
        leaves = {k: T(sc[v]).requires_grad_(True) for k, v in names.items()}
        for c in cams:
            render(c, sink=bucket.sink(), leaves=leaves)
        torch.cuda.synchronize()
        got = {k: bucket.alias[k] for k in names}
        for k in ("opacities", "colors"):
            e = float((got[k].reshape(want[k].shape) - want[k]).norm() / want[k].norm().clamp_min(1e-30))
            res["sink_%s_%s" % ("rows" if row_major else "planar", k)] = e
            assert e < GRAD_TOL or not finite, ("sink", row_major, k, e)
        for k in ("means3D", "scales", "rotations"):      # behind the chain rule: reported (conditioning: see run_case)
            e = float((got[k].reshape(want[k].shape) - want[k]).norm() / want[k].norm().clamp_min(1e-30))
            res["sink_%s_%s" % ("rows" if row_major else "planar", k)] = e
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=None, help="replay one case seed")
    ap.add_argument("--mode", choices=["oracle", "props"], default="oracle",
                    help="oracle: HIP against the CPU oracle; props: bands / bucket accumulation against the whole view (no oracle)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fuzz_parity.jsonl"))
    ap.add_argument("--seconds", type=float, default=1e9, help="stop drawing new cases after this long")
    args = ap.parse_args()
    import torch
    assert torch.cuda.is_available(), "needs the GPU"
    from oracle import oracle
    oracle.build()
    seeds = [args.only] if args.only is not None else [args.seed * 100000 + i for i in range(args.cases)]
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    t0, failed, done = time.time(), [], 0
    with open(args.out, "w") as f:
        for s in seeds:
            if time.time() - t0 > args.seconds:
                break
            try:
                r = run_case(oracle, s) if args.mode == "oracle" else run_props(s)
                r["ok"] = True
            except Exception as e:                            # noqa: BLE001 -- a fuzz harness reports everything
                desc = draw_case(s)[0]
                r = dict(desc, ok=False, error=repr(e)[:600], trace=traceback.format_exc()[-1500:])
                failed.append(s)
                print("FAIL", json.dumps(r, default=str)[:1500], flush=True)
            f.write(json.dumps(r, default=str) + "\n")
            f.flush()
            done += 1
    print("fuzz_parity: %d cases, %d failed %s in %.0f s" % (done, len(failed), failed[:20], time.time() - t0))
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()

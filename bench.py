#!/usr/bin/env python3
"""bench.py -- Gaussians/s, forward+backward at 1080p, of the MI355X rasterizer hot path.

  python bench.py --gpus 1 --steps K --warmup W                    (single GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W    (one rank per GPU, RCCL)

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): 1,000,000 random Gaussians (seed 0,
opacity 0.999, scales U(0, 0.5 N^-1/3)), 1920x1080, 8 orbit cameras PER GPU, through the drop-in
``diff_gaussian_rasterization_wodilate`` package (5-tuple flavour), loss = sum(image * w), backward to all
Gaussian attributes + means2D.  A "step" = every rank renders its 8 views forward+backward, gradients
are added by the backward kernels into one flat buffer per stream, summed per rank, and (N > 1) one
reduce-scatter + all-gather sums them across ranks (view-sharded data parallelism, weak scaling: per-GPU work
is fixed).  --streams S (default 3) views are in flight per GPU on S HIP streams.  Inputs are resident in HBM before
the timed region; the timed region contains no host synchronisation (tile-instance capacity comes from the
warm-up, overflow is verified afterwards).

Prints ONE JSON line on rank 0 (contract: see the task statement), including
  roofline     : dominant kernel, algorithmic bytes/launch / average launch duration (HIP events on the
                 launch stream; with S > 1 from a single-stream step right after the timed region, the
                 time-shared durations are in roofline_timed_region) vs the 8 TB/s HBM3E peak;
  cpu_baseline : the CPU oracle (oracle/, OpenMP, all host cores) timed on whole views of the same workload (~12 s).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--views", type=int, default=8, help="views per GPU per step")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--opacity", type=float, default=0.999, help="<0: random opacities")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("LOGRAST_BENCH_STREAMS", "3")),
                    help="independent views in flight per GPU (one HIP stream each)")
    ap.add_argument("--no-fused-accumulate", action="store_true",
                    help="let autograd accumulate each view's gradients (5 extra passes per view) instead of the "
                         "rasterizer adding them straight into the step's gradient bucket")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


def algorithmic_bytes(N, V, I, Px):
    """Per-view algorithmic HBM bytes per kernel (each datum counted once per producing /
    consuming stage, fp32, no implementation overhead)."""
    return {
        "project": 56 * N + 4 * N + 40 * V,
        "fill_keys": 8 * I,
        "sort": 8 * I + 4 * I,
        "blend_fwd": 44 * I + 28 * Px + 4 * V,
        "blend_bwd": 44 * I + 20 * Px + 36 * V,
        "project_bwd": 56 * N + 36 * V + 68 * N,
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback in the product path)"
    # LOGRAST_DIST_BACKEND=gloo + LOGRAST_SHARE_GPU=1: diagnostics only -- lets several ranks share one GPU (RCCL
    # refuses that) to exercise the multi-process path on a single-GPU box; the driver's runs use RCCL, one GPU each.
    backend = os.environ.get("LOGRAST_DIST_BACKEND", "nccl")
    share = os.environ.get("LOGRAST_SHARE_GPU", "0") == "1"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count() if share else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer
    from log_amd import _lib, rasterizer as R, scenes
    from log_amd.dist import GradientBucket

    N, W, H = args.gaussians, args.width, args.height
    Px = W * H
    sc = scenes.random_scene(N, seed=0, opacity=(None if args.opacity < 0 else args.opacity))
    # this rank's cameras: world*views angles on the orbit, round-robin
    all_cams = scenes.orbit_cameras(args.views * world, W=W, H=H, focal=2139.0 * W / 1920.0,
                                    end_deg=360.0 * (1 - 1.0 / (args.views * world)))
    cams = [all_cams[i] for i in range(rank, len(all_cams), world)]
    T = lambda a, g=False: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev, requires_grad=g)
    base = dict(means3D=T(sc["xyz"]), scales=T(sc["scaling"]), rotations=T(sc["rotation"]),
                opacities=T(sc["opacity"]), colors=T(sc["colors"]))
    # Views are independent until their gradients meet, so S of them are kept in flight on S HIP streams: the
    # atomic-bound binning kernels of one view overlap the ALU-bound compositing of another.  Every stream owns
    # leaf aliases of the (shared, read-only) attributes and its own flat gradient bucket; the buckets are
    # summed on the main stream at the end of the step, then reduced across ranks.
    S = max(1, min(args.streams, args.views))
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    lanes = []
    for _ in range(S):
        leaves = {k: v.detach().requires_grad_(True) for k, v in base.items()}
        bucket = GradientBucket(N, dev, world)
        bucket.attach(leaves)
        lanes.append((leaves, bucket))
    bucket = lanes[0][1]
    bg = T([1.0, 1.0, 1.0])
    wloss = torch.tensor(np.random.default_rng(1).random((3, H, W), dtype=np.float32), device=dev)
    rasts = []
    for cam in cams:
        rs = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam["FoVx"] * 0.5), tanfovy=math.tan(cam["FoVy"] * 0.5),
            bg=bg, scale_modifier=1.0, viewmatrix=T(cam["world_view_transform"]),
            projmatrix=T(cam["full_proj_transform"]), sh_degree=0, campos=T(cam["camera_center"]),
            prefiltered=False, debug=False)
        rasts.append(GaussianRasterizer(raster_settings=rs))

    def one_view(rast, leaves):
        means2D = torch.zeros(N, 3, device=dev, requires_grad=True)
        out = rast(means3D=leaves["means3D"], means2D=means2D, shs=None, colors_precomp=leaves["colors"],
                   opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
                   cov3D_precomp=None)
        # loss = sum(image * w): its gradient dL/dimage = w seeds the rasterizer's backward directly (the scalar
        # itself is consumed by nobody, so no reduction kernel is launched for it)
        out[0].backward(gradient=wloss)
        return out

    fused = not args.no_fused_accumulate

    def step():
        main = torch.cuda.current_stream(dev)
        for st in streams:
            st.wait_stream(main)
        for li, (leaves, bk) in enumerate(lanes):
            with torch.cuda.stream(streams[li]):
                bk.zero()
                if fused:
                    with R.accumulate_grads_into(bk.views):
                        for rast in rasts[li::S]:
                            one_view(rast, leaves)
                else:
                    for rast in rasts[li::S]:
                        one_view(rast, leaves)
        for st in streams:
            main.wait_stream(st)
        for _, bk in lanes[1:]:
            bucket.flat.add_(bk.flat)
        bucket.reduce()

    # ---- measure V and I per view once (exact mode: one 4-byte read-back per view) ----
    stats = []
    for rast in rasts:
        out = one_view(rast, lanes[0][0])
        n_inst, over, max_len, n_rect = R.last_state_info()
        stats.append((int((out[1] > 0).sum().item()), n_inst, max_len, n_rect))
        assert not over
    V = float(np.mean([s[0] for s in stats]))
    I = float(np.mean([s[1] for s in stats]))        # binned tile instances (after the support cull): what the kernels move
    I_rect = float(np.mean([s[3] for s in stats]))   # the reference's rect rule (its num_rendered)
    cap = int(max(s[1] for s in stats) * 1.02) + 1024
    # from here on: no host sync inside forward(); the longest tile list (it picks the sort's multi-block levels)
    # comes from the same measurement, with the same margin
    R.set_instance_capacity(cap, max_tile_len=int(max(s[2] for s in stats) * 1.02) + 64)
    R.overflow_since_reset(dev)         # clear the status block: from here on every forward is recorded in it

    for _ in range(args.warmup):
        step()
    timing = not args.no_kernel_timing
    if timing:
        _lib.profile_reset()
        _lib.profile_enable(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enqueued = time.perf_counter() - t0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if timing:
        _lib.profile_enable(False)
        prof_timed = _lib.profile_read()
        prof_serial = None
        if S > 1:
            # With several views in flight the kernels time-share the chip, so their HIP-event durations in the
            # timed region measure sharing, not the kernel.  One extra, untimed, single-stream step gives the
            # per-kernel durations that profiles/ (rocprofv3, serialized) can be compared with.
            _lib.profile_reset()
            _lib.profile_enable(True)
            lv0, bk0 = lanes[0]
            bk0.zero()
            with R.accumulate_grads_into(bk0.views):
                for rast in rasts:
                    one_view(rast, lv0)
            torch.cuda.synchronize()
            _lib.profile_enable(False)
            prof_serial = _lib.profile_read()
    chk = R.overflow_since_reset(dev)   # every forward since the capacity was set, on all streams
    assert not chk["overflowed"] and chk["max_instances"] <= cap, \
        "tile-instance capacity overflow inside the timed region: result invalid (%r)" % (chk,)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_gaussians = float(N) * args.views * world * args.steps
    value = total_gaussians / elapsed

    result = {
        "metric": "Gaussians/sec fwd+bwd @1080p", "value": value, "unit": "Gaussians/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "C2: %d random Gaussians (seed 0, opacity %s), %dx%d, %d orbit views per GPU, "
                        "fwd+bwd of sum(image*w), wodilate (5-tuple) flavour" %
                        (N, "rand" if args.opacity < 0 else args.opacity, W, H, args.views),
            "gaussians": N, "width": W, "height": H, "views_per_gpu": args.views,
            "visible_per_view": V, "tile_instances_per_view": I, "tile_instances_per_view_reference_rect_rule": I_rect,
            "streams_per_gpu": S, "fused_gradient_accumulation": fused,
            "parallelism": ("view-sharded dp%d, 1 reduce-scatter+all-gather of %d floats/step" %
                            (world, bucket.flat.numel()) if world > 1 else "single GPU") +
                           ", %d views in flight per GPU (HIP streams)" % S,
        },
        "ms_per_view": 1e3 * elapsed / (args.steps * args.views),
        "host_enqueue_ms_per_view": 1e3 * t_enqueued / (args.steps * args.views),
    }

    if rank == 0:
        alg = algorithmic_bytes(N, V, I, Px)
        total_alg = 184 * N + 116 * V + 108 * I + 48 * Px
        result["algorithmic_GBs_whole_view"] = total_alg / (elapsed / (args.steps * args.views)) / 1e9
        # the same box's device-copy bandwidth (the measured roof next to the 8 TB/s spec figure)
        copy_gbs = measured_copy_bandwidth(dev)
        result["measured_copy_GBs"] = copy_gbs
        result["algorithmic_frac_of_measured_copy"] = result["algorithmic_GBs_whole_view"] / copy_gbs
        if timing:
            def roof(prof):
                kern = {}
                for name, (ms, cnt) in prof.items():
                    kern[name] = {"avg_us": 1e3 * ms / cnt, "launches": int(cnt)}
                    if name in alg:   # per-kernel algorithmic GB/s against the same 8 TB/s roof
                        gbs = alg[name] / (1e-3 * ms / cnt) / 1e9
                        kern[name].update(alg_GBs=gbs, hbm_frac=gbs / HBM_PEAK_GBS)
                sort_ms = sum(prof[k][0] for k in prof if k.startswith("sort"))
                merged = {k: prof[k][0] for k in prof if not k.startswith("sort")}
                if sort_ms:
                    merged["sort"] = sort_ms
                dom = max(merged, key=merged.get)
                launches = prof[dom][1] if dom in prof else prof["sort_small"][1]
                avg_s = merged[dom] / launches / 1e3
                achieved = alg.get(dom, 0) / avg_s / 1e9
                return kern, {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                              "traffic": pmc_traffic(dom, N, W, H),
                              # the compositing kernels are bound by fp32 VALU issue, which the hbm/mfma vocabulary of
                              # this object cannot name: fraction of SIMD issue cycles spent in VALU ops (PMC pass)
                              "valu_issue_frac": pmc_traffic(dom, N, W, H, "valu_active_frac_at_2p4GHz"),
                              "algorithmic_bytes_per_launch": alg.get(dom, 0), "avg_launch_us": avg_s * 1e6}
            if prof_serial is None:
                result["kernels"], result["roofline"] = roof(prof_timed)
                result["roofline"]["measured"] = "HIP events on the launch stream inside the timed region (one stream)"
            else:
                # S > 1: the kernel's own duration comes from the single-stream step (this is what rocprofv3, which
                # serialises kernels, reports for the same command); the time-shared durations of the timed region
                # are kept next to it.
                result["kernels"], result["roofline"] = roof(prof_serial)
                result["roofline"]["measured"] = ("HIP events on the launch stream, single-stream step run right after "
                                                  "the timed region (inside it %d views are in flight and kernels "
                                                  "time-share the chip: see roofline_timed_region)" % S)
                result["kernels_timed_region"], result["roofline_timed_region"] = roof(prof_timed)
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only: the host cores are shared by all ranks
            result["cpu_baseline"] = cpu_baseline(sc, cams, wloss.cpu().numpy(), N)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_copy_bandwidth(dev, mib=1024, reps=5):
    """Device-to-device copy of `mib` MiB (read + write counted), best of `reps`: GB/s."""
    a = torch.empty(mib * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    del a, b
    return 2 * mib * 1024 * 1024 / (best * 1e-3) / 1e9


def pmc_traffic(kernel, N, W, H, field="traffic_bytes"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r*_traffic.json,
    FETCH_SIZE/WRITE_SIZE collected and corrected as MI355X_MICROARCH.md prescribes); None when no profile of this
    exact workload is committed -- counters cannot be collected from inside the timed run."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            wl = d.get("workload", {})
            if (wl.get("gaussians"), wl.get("width"), wl.get("height")) == (N, W, H) and kernel in d["kernels"]:
                return d["kernels"][kernel][field]
        except (OSError, ValueError, KeyError):
            continue
    return None


def cpu_baseline(sc, cams, wloss, N, budget_s=12.0, max_passes=64):
    """The CPU oracle (test infrastructure; OpenMP over all host cores) on a bounded sample of the same
    workload: whole views (all N Gaussians, forward+backward), cycling through this rank's cameras until
    ~budget_s seconds of CPU work have been spent."""
    from oracle import oracle
    cores = os.cpu_count() or 1
    oracle.lib()
    views = []
    for cam in cams:
        tfx, tfy = math.tan(cam["FoVx"] * 0.5), math.tan(cam["FoVy"] * 0.5)
        views.append(oracle.make_view(cam["image_width"], cam["image_height"], tfx, tfy,
                                      cam["world_view_transform"], cam["full_proj_transform"], [1, 1, 1]))
    passes, t0 = 0, time.perf_counter()
    while passes < max_passes and (passes == 0 or time.perf_counter() - t0 < budget_s):
        v = views[passes % len(views)]
        f = oracle.forward(v, sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["colors"])
        oracle.backward(v, f, wloss)
        passes += 1
    dt = time.perf_counter() - t0
    return {"value": N * passes / dt, "unit": "Gaussians/s", "cores": cores, "kind": "port",
            "sample": "%d view passes (cycling the 8 views), all %d Gaussians each, forward+backward, %.1f s total"
                      % (passes, N, dt)}


if __name__ == "__main__":
    main()

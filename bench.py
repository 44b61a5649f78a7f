#!/usr/bin/env python3
"""bench.py -- Gaussians/s, forward+backward at 1080p, of the MI355X rasterizer hot path.

  python bench.py --gpus 1 --steps K --warmup W                    (single GPU)
  python bench.py --gpus N ...                                     (re-launches itself under torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W    (one rank per GPU, RCCL)

Headline workload = the north-star point of BASELINE.json (configs[3] per GPU; SURVEY.md 8d "throughput target"):
30,000,000 random Gaussians (seed 0, opacity 0.999, scales U(0, 0.5 N^-1/3)), 1920x1080, 8 orbit cameras PER GPU,
through the drop-in ``diff_gaussian_rasterization_wodilate`` package (5-tuple flavour), loss = sum(image * w),
backward to all Gaussian attributes + means2D.  A "step" = every rank renders its 8 views forward+backward, gradients are
added by the backward kernels into one flat buffer per stream, summed per rank, and (N > 1) summed across ranks by
reduce-scatter + all-gather (view-sharded data parallelism, weak scaling: per-GPU work is fixed; the views of a step go in
--exchange-parts groups and a group's reduce-scatter runs on a side stream under the next group's rendering,
log_amd.dist.StepExchange).  Inputs are
resident in HBM before the timed region; the timed region contains no host synchronisation (tile-instance capacity
comes from the warm-up; every forward records itself in the rasterizer's status block, checked afterwards).

Prints ONE JSON line on rank 0 (contract: see the task statement), including
  value        : the pipelined mode above (what a multi-view training step of this framework runs);
  modes        : the same workload also in the DROP-IN DEFAULT mode -- one stream, exact buffer sizing (one 4-byte
                 read-back per forward, like the third-party package's num_rendered), gradients accumulated by autograd;
  roofline     : dominant kernel, algorithmic bytes/launch / average launch duration (HIP events on the launch stream,
                 from a single-stream step when several views are in flight) vs the 8 TB/s HBM3E peak; `traffic` from the
                 committed PMC passes at this workload (profiles/r*_traffic*.json);
  cpu_baseline : the CPU oracle (oracle/, OpenMP, all host cores) timed on whole views of the same workload;
  secondary    : (N = 1 only) C2 = configs[1] (1 M Gaussians, same harness, both modes) and C3 = configs[2] (10 M-point
                 LoD tree, SH degree 3, level selection on: one LoG training view end to end through the drop-ins).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gaussians", type=int, default=30_000_000)
    ap.add_argument("--views", type=int, default=8, help="views per GPU per step")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--opacity", type=float, default=0.999, help="<0: random opacities")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("LOGRAST_BENCH_STREAMS", "0")),
                    help="independent views in flight per GPU, one HIP stream each (0 = auto: 3 up to 4 M Gaussians, where "
                         "the binning kernels leave CUs idle; 1 beyond, where every kernel fills the chip)")
    ap.add_argument("--no-fused-accumulate", action="store_true",
                    help="let autograd accumulate each view's gradients (5 extra passes per view) instead of the "
                         "rasterizer adding them straight into the step's gradient bucket")
    ap.add_argument("--no-graphs", action="store_true",
                    help="pipelined mode: enqueue every view's ~15 launches from Python instead of replaying one captured "
                         "HIP graph per view")
    ap.add_argument("--exchange-parts", type=int, default=int(os.environ.get("LOGRAST_EXCHANGE_PARTS", "4")),
                    help="N > 1: groups of views per step, each reduce-scattered under the next group's rendering")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C2 / C3 legs (they run at N = 1 only)")
    ap.add_argument("--no-dropin-mode", action="store_true", help="skip the drop-in-default measurement of the headline")
    ap.add_argument("--c3-torch", action="store_true", help="C3 leg: also time the reference-style torch pipeline")
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


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: re-run under torch.distributed.run, one rank per GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class RasterWorkload:
    """N random Gaussians, `views` orbit cameras for this rank, the drop-in rasterizer objects, loss weights."""

    def __init__(self, args, N, dev, rank, world, torch, np):
        from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer
        from log_amd import scenes
        self.N, self.W, self.H, self.dev, self.views = N, args.width, args.height, dev, args.views
        W, H = self.W, self.H
        self.sc = scenes.random_scene(N, seed=0, opacity=(None if args.opacity < 0 else args.opacity))
        all_cams = scenes.orbit_cameras(args.views * world, W=W, H=H, focal=2139.0 * W / 1920.0,
                                        end_deg=360.0 * (1 - 1.0 / (args.views * world)))
        self.cams = [all_cams[i] for i in range(rank, len(all_cams), world)]   # round-robin view ownership
        T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
        self.base = dict(means3D=T(self.sc["xyz"]), scales=T(self.sc["scaling"]), rotations=T(self.sc["rotation"]),
                         opacities=T(self.sc["opacity"]), colors=T(self.sc["colors"]))
        bg = T([1.0, 1.0, 1.0])
        self.wloss = torch.tensor(np.random.default_rng(1).random((3, H, W), dtype=np.float32), device=dev)
        self.rasts = []
        for cam in self.cams:
            rs = GaussianRasterizationSettings(
                image_height=H, image_width=W, tanfovx=math.tan(cam["FoVx"] * 0.5), tanfovy=math.tan(cam["FoVy"] * 0.5),
                bg=bg, scale_modifier=1.0, viewmatrix=T(cam["world_view_transform"]),
                projmatrix=T(cam["full_proj_transform"]), sh_degree=0, campos=T(cam["camera_center"]),
                prefiltered=False, debug=False)
            self.rasts.append(GaussianRasterizer(raster_settings=rs))
        self.torch = torch
        self.zero_means2d = True

    def one_view(self, rast, leaves):
        torch = self.torch
        # means2D is the third-party API's placeholder that receives dL/dmeans2D; no kernel reads its values.  LoG's
        # renderer.py zero-fills a fresh one per view (the drop-in default mode does the same); the pipelined step
        # allocates it uninitialised (30 M Gaussians: a 360 MB memset per view less).
        means2D = (torch.zeros if self.zero_means2d else torch.empty)(self.N, 3, device=self.dev).requires_grad_(True)
        out = rast(means3D=leaves["means3D"], means2D=means2D, shs=None, colors_precomp=leaves["colors"],
                   opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
                   cov3D_precomp=None)
        # loss = sum(image * w): its gradient dL/dimage = w seeds the rasterizer's backward directly (the scalar
        # itself is consumed by nobody, so no reduction kernel is launched for it)
        out[0].backward(gradient=self.wloss)
        return out


def measure(args, wl, S, fused, steps, warmup, world, timing, sync_free=True, graphs=False):
    """Times `steps` steps of workload `wl` with S views in flight.  sync_free=False + S=1 + fused=False is the drop-in
    default mode.  graphs: every view (forward + backward, a fixed sequence of ~15 launches once the capacity is fixed)
    is captured into one HIP graph per (stream, camera) after the warm-up and replayed in the timed region.
    -> dict(elapsed, t_enqueued, V, I, I_rect, prof_timed, prof_serial, bucket_floats, graphs)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from log_amd import _lib, rasterizer as R
    from log_amd.dist import GradientBucket, StepExchange
    dev, N = wl.dev, wl.N
    wl.zero_means2d = not (sync_free and fused)
    rank = dist.get_rank() if world > 1 else 0
    parts = max(1, min(int(args.exchange_parts), len(wl.rasts) // S)) if world > 1 else 1
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    # The step's gradient exchange (log_amd.dist.StepExchange): the rank's views in `parts` consecutive groups with a
    # bucket each; group g's reduce-scatter runs on a side stream under the rendering of group g + 1.
    ex = StepExchange(N, dev, world, rank, parts=parts, track_seen=False)   # (the gradient sum only: no optimizer here)
    lanes = []
    for li in range(S):   # every stream: leaf aliases of the (shared, read-only) attributes + its own flat gradient buckets
        leaves = {k: v.detach().requires_grad_(True) for k, v in wl.base.items()}
        bks = ex.buckets if li == 0 else [GradientBucket(N, dev, world, track_seen=False) for _ in range(parts)]
        bks[0].attach(leaves)
        lanes.append((leaves, bks))
    lane_views = [wl.rasts[li::S] for li in range(S)]
    part_of = [[min(j * parts // max(len(lv), 1), parts - 1) for j in range(len(lv))] for lv in lane_views]

    lane_graphs = None

    def step():
        main = torch.cuda.current_stream(dev)
        for st in streams:
            st.wait_stream(main)
        for li, (_, bks) in enumerate(lanes):
            with torch.cuda.stream(streams[li]):
                for bk in bks:
                    bk.zero()
        for part in range(parts):
            for li, (leaves, bks) in enumerate(lanes):
                mine = [j for j in range(len(lane_views[li])) if part_of[li][j] == part]
                with torch.cuda.stream(streams[li]):
                    if lane_graphs is not None:
                        for j in mine:
                            lane_graphs[li][j].replay()
                    elif fused:
                        with R.accumulate_grads_into(bks[part].views):
                            for j in mine:
                                wl.one_view(lane_views[li][j], leaves)
                    else:
                        if parts > 1:
                            bks[part].attach(leaves)
                        for j in mine:
                            wl.one_view(lane_views[li][j], leaves)
            for st in streams:
                main.wait_stream(st)                    # (a dependency on the device, not a host wait)
            for _, bks in lanes[1:]:
                ex.buckets[part].flat.add_(bks[part].flat)
            if world > 1:
                ex.launch(part)
        if world > 1:
            ex.all_gather_grads(ex.finish())            # every rank ends the step with the whole gradient sum

    # ---- V and I per view, measured once in exact mode (one 4-byte read-back per view) ----
    R.set_instance_capacity(None)
    stats = []
    for rast in wl.rasts:
        out = wl.one_view(rast, lanes[0][0])
        n_inst, over, max_len, n_rect = R.last_state_info(dev)
        stats.append((int((out[1] > 0).sum().item()), n_inst, max_len, n_rect))
        assert not over
        del out
    res = {"V": float(np.mean([s[0] for s in stats])), "I": float(np.mean([s[1] for s in stats])),
           "I_rect": float(np.mean([s[3] for s in stats])), "bucket_floats": int(ex.buckets[0].flat.numel()),
           "exchange_parts": parts}
    cap = int(max(s[1] for s in stats) * 1.02) + 1024
    if sync_free:
        # from here on: no host sync inside forward(); the longest tile list (it picks the sort's multi-block levels)
        # comes from the same measurement, with the same margin
        R.set_instance_capacity(cap, max_tile_len=int(max(s[2] for s in stats) * 1.02) + 64)
    R.overflow_since_reset(dev)         # clear the status block: from here on every forward is recorded in it
    for _ in range(warmup):
        step()
    res["graphs"] = False
    if graphs and sync_free and fused:
        # One HIP graph per (stream, camera): the view's forward and backward -- memset, projection, scan, fill, sorts,
        # compositing, reverse walk, chain rule, plus the two torch kernels around them -- replayed with one call.
        # Buffers come from a private pool per stream (views of different streams run concurrently).
        try:
            torch.cuda.synchronize()
            built = []
            for li, (leaves, bks) in enumerate(lanes):
                pool, gl = torch.cuda.graph_pool_handle(), []
                for j, rast in enumerate(lane_views[li]):
                    with R.accumulate_grads_into(bks[part_of[li][j]].views):
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, pool=pool, stream=streams[li]):
                            wl.one_view(rast, leaves)
                    gl.append(g)
                built.append(gl)
            torch.cuda.synchronize()
            lane_graphs = built
            step()                      # one replayed step before the clock starts
            res["graphs"] = True
        except Exception as e:          # capture is an optimisation of the enqueue path only: say so and go on eagerly
            lane_graphs = None
            res["graphs_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
            torch.cuda.synchronize()
    kernel_timing = timing and lane_graphs is None      # HIP events are not recorded inside a replayed graph
    if kernel_timing:
        _lib.profile_reset()
        _lib.profile_enable(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    res["t_enqueued"] = time.perf_counter() - t0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    res["elapsed"] = time.perf_counter() - t0
    res["prof_timed"] = res["prof_serial"] = None
    if timing:
        if kernel_timing:
            _lib.profile_enable(False)
            res["prof_timed"] = _lib.profile_read()
        if S > 1 or not kernel_timing:
            # With several views in flight the kernels time-share the chip, so their HIP-event durations in the
            # timed region measure sharing, not the kernel.  One extra, untimed, single-stream step gives the
            # per-kernel durations that profiles/ (rocprofv3, serialized) can be compared with.
            _lib.profile_reset()
            _lib.profile_enable(True)
            lv0, bk0 = lanes[0][0], lanes[0][1][0]
            bk0.zero()
            with R.accumulate_grads_into(bk0.views):
                for rast in wl.rasts:
                    wl.one_view(rast, lv0)
            torch.cuda.synchronize()
            _lib.profile_enable(False)
            res["prof_serial"] = _lib.profile_read()
    chk = R.overflow_since_reset(dev)   # every forward since the capacity was set, on all streams
    assert not chk["overflowed"] and chk["max_instances"] <= cap, \
        "tile-instance capacity overflow inside the timed region: result invalid (%r)" % (chk,)
    R.set_instance_capacity(None)
    wl.zero_means2d = True
    if world > 1:
        t = torch.tensor([res["elapsed"]], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res["elapsed"] = float(t.item())
    del lane_graphs, lanes, ex
    return res


def mode_summary(N, views, world, steps, r):
    return {"ms_per_view": 1e3 * r["elapsed"] / (steps * views), "ms_per_step": 1e3 * r["elapsed"] / steps,
            "value": float(N) * views * world * steps / r["elapsed"],
            "host_enqueue_ms_per_view": 1e3 * r["t_enqueued"] / (steps * views)}


def auto_streams(args, N):
    S = args.streams if args.streams > 0 else (3 if N <= 4_000_000 else 1)
    return max(1, min(S, args.views))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args))
    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
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
    n_ranks = dist.get_world_size() if world > 1 else 1     # what the process group really holds

    N, W, H = args.gaussians, args.width, args.height
    Px = W * H
    timing = not args.no_kernel_timing
    fused = not args.no_fused_accumulate
    wl = RasterWorkload(args, N, dev, rank, world, torch, np)
    S = auto_streams(args, N)
    r = measure(args, wl, S, fused, args.steps, args.warmup, world, timing, graphs=not args.no_graphs)
    V, I, I_rect, elapsed = r["V"], r["I"], r["I_rect"], r["elapsed"]
    head = mode_summary(N, args.views, world, args.steps, r)

    result = {
        "metric": "Gaussians/sec fwd+bwd @1080p", "value": head["value"], "unit": "Gaussians/s", "n_gpus": n_ranks,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "%s: %d random Gaussians (seed 0, opacity %s), %dx%d, %d orbit views per GPU, "
                        "fwd+bwd of sum(image*w), wodilate (5-tuple) flavour" %
                        ("north-star point (BASELINE.json north_star / configs[3] per GPU)" if N >= 30_000_000 else
                         ("C2 (configs[1])" if N == 1_000_000 else "custom"),
                         N, "rand" if args.opacity < 0 else args.opacity, W, H, args.views),
            "gaussians": N, "width": W, "height": H, "views_per_gpu": args.views,
            "visible_per_view": V, "tile_instances_per_view": I, "tile_instances_per_view_reference_rect_rule": I_rect,
            "streams_per_gpu": S, "fused_gradient_accumulation": fused,
            "parallelism": ("view-sharded dp%d, reduce-scatter of %d floats/step in %d groups of views (each under the next "
                            "group's rendering, side stream) + one all-gather" %
                            (n_ranks, r["bucket_floats"], r["exchange_parts"]) if world > 1 else "single GPU") +
                           ", %d view(s) in flight per GPU (HIP streams)" % S,
        },
        "ms_per_view": head["ms_per_view"], "host_enqueue_ms_per_view": head["host_enqueue_ms_per_view"],
        "modes": {"pipelined": dict(head, streams=S, sync_free=True, fused_gradient_accumulation=fused, hip_graphs=r["graphs"],
                                    note="value of this line: capacity from the warm-up, no host sync in forward(), "
                                         "gradients added by the backward kernels into the step's bucket")},
    }
    if not args.no_dropin_mode:
        # the drop-in default: what LoG's unmodified renderer.py gets -- one stream, one 4-byte read-back per forward,
        # autograd accumulating every view's gradients
        dsteps = max(1, min(args.steps, 3))
        rd = measure(args, wl, 1, False, dsteps, 1, world, False, sync_free=False)
        result["modes"]["dropin_default"] = dict(
            mode_summary(N, args.views, world, dsteps, rd), streams=1, sync_free=False,
            fused_gradient_accumulation=False, steps=dsteps,
            note="one stream, exact buffer sizing (4-byte read-back per forward), autograd accumulation")

    if rank == 0:
        alg = algorithmic_bytes(N, V, I, Px)
        total_alg = 184 * N + 116 * V + 108 * I + 48 * Px
        result["algorithmic_bytes_per_view"] = total_alg
        result["algorithmic_GBs_whole_view"] = total_alg / (elapsed / (args.steps * args.views)) / 1e9
        # the same box's device-copy bandwidth (the measured roof next to the 8 TB/s spec figure)
        copy_gbs = measured_copy_bandwidth(dev)
        result["measured_copy_GBs"] = copy_gbs
        result["algorithmic_frac_of_measured_copy"] = result["algorithmic_GBs_whole_view"] / copy_gbs
        if r.get("graphs_error"):
            result["graphs_error"] = r["graphs_error"]
        if timing:
            if r["prof_serial"] is None:
                result["kernels"], result["roofline"] = roof(r["prof_timed"], alg, N, W, H)
                result["roofline"]["measured"] = "HIP events on the launch stream inside the timed region (one stream)"
            elif r["prof_timed"] is None:
                result["kernels"], result["roofline"] = roof(r["prof_serial"], alg, N, W, H)
                result["roofline"]["measured"] = ("HIP events on the launch stream, one eagerly launched single-stream step "
                                                  "run right after the timed region (inside it the views are replayed "
                                                  "HIP graphs, which carry no events)")
            else:
                # S > 1: the kernel's own duration comes from the single-stream step (this is what rocprofv3, which
                # serialises kernels, reports for the same command); the time-shared durations of the timed region
                # are kept next to it.
                result["kernels"], result["roofline"] = roof(r["prof_serial"], alg, N, W, H)
                result["roofline"]["measured"] = ("HIP events on the launch stream, single-stream step run right after "
                                                  "the timed region (inside it %d views are in flight and kernels "
                                                  "time-share the chip: see roofline_timed_region)" % S)
                result["kernels_timed_region"], result["roofline_timed_region"] = roof(r["prof_timed"], alg, N, W, H)
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only: the host cores are shared by all ranks
            result["cpu_baseline"] = cpu_baseline(wl.sc, wl.cams, wl.wloss.cpu().numpy(), N)
    del wl
    torch.cuda.empty_cache()

    if world == 1 and not args.no_secondary:
        sec = {}
        try:
            if N != 1_000_000:
                wl2 = RasterWorkload(args, 1_000_000, dev, rank, world, torch, np)
                S2, steps2 = auto_streams(args, 1_000_000), max(args.steps, 20)   # (a 0.65 ms view: 20 steps = 0.1 s)
                r2 = measure(args, wl2, S2, fused, steps2, max(args.warmup, 2), world, timing, graphs=not args.no_graphs)
                c2 = {"workload": "C2 (BASELINE.json configs[1]): 1000000 random Gaussians (seed 0, opacity %s), %dx%d, "
                                  "%d orbit views, same harness as the headline"
                                  % ("rand" if args.opacity < 0 else args.opacity, W, H, args.views),
                      "visible_per_view": r2["V"], "tile_instances_per_view": r2["I"],
                      "modes": {"pipelined": dict(mode_summary(1_000_000, args.views, 1, steps2, r2), streams=S2,
                                                  hip_graphs=r2["graphs"])}}
                if timing:
                    alg2 = algorithmic_bytes(1_000_000, r2["V"], r2["I"], Px)
                    c2["kernels"], c2["roofline"] = roof(r2["prof_serial"] or r2["prof_timed"], alg2, 1_000_000, W, H)
                if not args.no_dropin_mode:
                    rd2 = measure(args, wl2, 1, False, 3, 1, world, False, sync_free=False)
                    c2["modes"]["dropin_default"] = dict(mode_summary(1_000_000, args.views, 1, 3, rd2), streams=1)
                sec["c2"] = c2
                del wl2
                torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_log_step
            sec["c3"] = bench_log_step.c3_pipeline(views=4, with_torch=args.c3_torch, dev=dev)
        except Exception as e:   # the headline stands on its own; say what happened to the rest
            sec["error"] = "%s: %s" % (type(e).__name__, e)
        result["secondary"] = sec

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def roof(prof, alg, N, W, H):
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


def measured_copy_bandwidth(dev, mib=1024, reps=5):
    """Device-to-device copy of `mib` MiB (read + write counted), best of `reps`: GB/s."""
    import torch
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
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r*_traffic*.json,
    FETCH_SIZE/WRITE_SIZE collected and corrected as MI355X_MICROARCH.md prescribes); None when no profile of this
    exact workload is committed -- counters cannot be collected from inside the timed run."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            wl = d.get("workload", {})
            if (wl.get("gaussians"), wl.get("width"), wl.get("height")) == (N, W, H) and kernel in d["kernels"]:
                return d["kernels"][kernel].get(field)
        except (OSError, ValueError, KeyError):
            continue
    return None


def cpu_baseline(sc, cams, wloss, N, budget_s=12.0, max_passes=64):
    """The CPU oracle (test infrastructure; OpenMP over all host cores) on a bounded sample of the same
    workload: whole views (all N Gaussians, forward+backward), cycling through this rank's cameras until
    ~budget_s seconds of CPU work have been spent (at least one pass)."""
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
            "sample": "%d view pass(es) (cycling the %d views), all %d Gaussians each, forward+backward, %.1f s total"
                      % (passes, len(views), N, dt)}


if __name__ == "__main__":
    main()

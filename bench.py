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
added by the backward kernels into one flat buffer per stream (row-major: one 64-byte row of running sums per Gaussian,
log_amd.dist.GradientBucket(row_major=True); --planar-bucket: five attribute-major arrays), summed per rank, and (N > 1) summed across ranks by
reduce-scatter + all-gather (view-sharded data parallelism, weak scaling: per-GPU work is fixed).  --exchange-parts G > 1
splits the step's views into G groups with a FULL-SIZE bucket each, group g's reduce-scatter running on a side stream under
group g + 1's rendering (log_amd.dist.StepExchange) -- which pays only when a group's exchange is smaller than the step's
(touched-block exchange of level-of-detail views); for this workload, where every view touches rows all over the model, the
last group's reduce-scatter is as large as the whole step's, so what stays exposed (one reduce-scatter + the all-gather) is
the same with G = 1, and G groups cost G x the link traffic and G x the bucket zero-fills: the default is 1.  --exchange auto (default) measures in the warm-up how many of a rank's rows its views touched and
takes the ROW-SPARSE exchange when that is less than half (the opaque headline: 24 %; only touched rows travel, packed per owner:
log_amd.dist.GradientBucket.reduce_scatter_rows_sparse + the sparse all-gather; device pack / unpack kernels), else the dense one.  Inputs are
resident in HBM before the timed region; the timed region contains no host synchronisation (tile-instance capacity
comes from the warm-up; every forward records itself in the rasterizer's status block, checked afterwards).

The LAST stdout line of rank 0 is the compact contract object (< 4 KB: compact_line()); the FULL result described below goes
to bench_full.json next to this file (and gpurun_out/bench_full.json; --print-full: also an earlier stdout line):
  value        : the pipelined mode above (what a multi-view training step of this framework runs);
  modes        : the same workload also in the DROP-IN DEFAULT mode -- one stream, the package's default forward (stage 2
                 enqueued speculatively, one read-back per forward on a side stream that the launch stream never waits
                 for; always exact), gradients added in place into the leaves' .grad;
  modes.pipelined_opacity_rand : (N = 1) the same 30 M point with opacity = rand(N) (SURVEY 8d asks for both variants): no
                 early termination to hide behind -- every list is walked to its end;
  roofline     : dominant kernel, algorithmic bytes/launch / average launch duration (HIP events on the launch stream,
                 from a single-stream step when several views are in flight) vs the 8 TB/s HBM3E peak.  Bytes are the
                 EFFECTIVE algorithmic bytes: SURVEY 8d's per-unit figures x the units the launch really processes (list
                 entries up to the deepest contributor of their tile instead of all I; Gaussians with point_weight > 0
                 instead of all V for the chain rule), so no fraction can exceed 1 by crediting bytes nobody has to move;
                 the plain SURVEY formula is kept beside it (`*_survey_formula`).  `traffic` comes from the committed PMC
                 passes at this workload (profiles/r*_traffic*.json: builder-collected, `traffic_source` says so);
  measured roofs: `measured_stream_copy_GBs` = this library's own float4 streaming-copy kernel (lograst_stream_copy) on the
                 same box in the same run, `measured_copy_GBs` = torch's Tensor.copy_ (what round 2 divided by);
  forward_only : torch.no_grad() rendering (what the reference times: CUDA events around renderer.vis in
                 apps/train.py:53-59,100-108): ms/view and fps for the 30 M point, C2 and C3;
  cpu_baseline : the CPU oracle (oracle/, OpenMP, all host cores) timed on whole views of the same workload;
  secondary    : (N = 1 only) C2 = configs[1] (1 M Gaussians, same harness, both modes), C3 = configs[2] (10 M-point
                 LoD tree, SH degree 3, level selection on: one LoG training view end to end through the drop-ins) and
                 c5_band = configs[4] on ONE of its 8 GPUs (100 M Gaussians, 3840x2160, one band of tile rows: pre-pass +
                 gather / scatter, clipped inside the projection with autograd gradients, and with the gradient sink).
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
BENCH_ROUND = 6        # committed counter profiles of EARLIER rounds are stale (the kernels changed): roofline.traffic drops them


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
    ap.add_argument("--planar-bucket", action="store_true",
                    help="fused accumulation into attribute-major buckets (five arrays) instead of one 64-byte row per Gaussian")
    ap.add_argument("--no-graphs", action="store_true",
                    help="pipelined mode: enqueue every view's ~15 launches from Python instead of replaying one captured "
                         "HIP graph per view")
    ap.add_argument("--exchange-parts", type=int, default=int(os.environ.get("LOGRAST_EXCHANGE_PARTS", "0")),
                    help="N > 1: groups of views per step, each exchanged under the next group's rendering.  0 (default) = one "
                         "group per view when the row-sparse exchange runs (STREAMED: a view's touched rows are packed, cleared "
                         "and sent while the next view renders; log_amd.dist.StepExchange), one group otherwise (a dense group "
                         "exchange is as large as the whole step's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-out", default=None,
                    help="where the full result object goes (default: bench_full.json next to bench.py and in gpurun_out/)")
    ap.add_argument("--print-full", action="store_true",
                    help="also print the full result object (what bench_full.json holds) on an EARLIER stdout line")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-one-rank-leg", action="store_true",
                    help="skip the leg that runs the N > 1 step through a one-rank RCCL group in a child process (N = 1 only)")
    ap.add_argument("--no-pack-hint", action="store_true",
                    help="streamed row-sparse exchange: scan the gradient rows themselves instead of the view's point_weight")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C2 / C3 legs (they run at N = 1 only)")
    ap.add_argument("--no-dropin-mode", action="store_true", help="skip the drop-in-default measurement of the headline")
    ap.add_argument("--c3-torch", action="store_true", help="C3 leg: also time the reference-style torch pipeline")
    ap.add_argument("--no-rand-variant", action="store_true", help="skip the opacity = rand variant of the headline (N = 1)")
    ap.add_argument("--no-forward-only", action="store_true", help="skip the torch.no_grad() legs")
    ap.add_argument("--no-c5-band", action="store_true", help="skip the C5 band leg (100 M Gaussians, 4K, one of 8 bands)")
    ap.add_argument("--no-trained-like", action="store_true",
                    help="skip the trained-like variant of the headline (log-normal scales, bounded anisotropy, sigmoid-normal opacity)")
    ap.add_argument("--scene", choices=("random", "trained"), default="random",
                    help="headline scene generator: check_gui's uniform draws (SURVEY 8d) or log_amd.scenes.trained_like_scene")
    ap.add_argument("--exchange", choices=("auto", "dense", "compact", "sparse"), default=os.environ.get("LOGRAST_EXCHANGE", "auto"),
                    help="N > 1: dense reduce-scatter + all-gather, touched-row-block exchange, row-sparse exchange (only rows "
                         "with a non-zero gradient travel), or decided once from the warm-up (auto: row-sparse when fewer than "
                         "half of a rank's rows are touched, else dense)")
    return ap.parse_args()


def algorithmic_bytes(N, V, I, Px, eff=None):
    """Per-view algorithmic HBM bytes per kernel (each datum counted once per producing /
    consuming stage, fp32, no implementation overhead).  eff = effective_units(): the same per-unit figures times the
    units the launch really has to process -- list entries up to the deepest contributor of their tile (forward: up to
    where the tile's last pixel stops) and, for the chain rule, the Gaussians some pixel composited."""
    alg = {
        "project": 56 * N + 4 * N + 40 * V,
        "fill_keys": 8 * I,
        "sort": 8 * I + 4 * I,
        "blend_fwd": 44 * I + 28 * Px + 4 * V,
        "blend_bwd": 44 * I + 20 * Px + 36 * V,
        "project_bwd": 56 * N + 36 * V + 68 * N,
    }
    if eff is None:
        return alg, dict(alg)
    e = dict(alg)
    e["blend_fwd"] = 44 * eff["I_walked_fwd"] + 28 * Px + 4 * eff["V_live"]
    e["blend_bwd"] = 44 * eff["I_walked_bwd"] + 20 * Px + 36 * eff["V_live"]
    e["project_bwd"] = 8 * N + (56 + 36 + 68) * eff["V_live"]      # live flags (radii, point_weight) of all, rows of the live
    return alg, e


def effective_units(wl):
    """Per view (mean over this rank's cameras), from one raw forward each: V_live = Gaussians with point_weight > 0 (the
    rows the chain rule processes), I_walked_bwd = sum over tiles of the deepest contributor's list position (where the
    reverse walk starts), I_walked_fwd = sum over tiles of how far the forward has to read: the whole list when some pixel
    of the tile never saturates (final_T >= 0.01: the stop needs T < 1e-4 / (1 - 0.99)), else the deepest contributor
    rounded up to the next 64-entry chunk (an estimate: the exact stop position is not an output)."""
    import torch
    from log_amd import rasterizer as R
    W, H, dev = wl.W, wl.H, wl.dev
    gx, gy = (W + 15) // 16, (H + 15) // 16
    acc = {"V_live": 0.0, "I_walked_fwd": 0.0, "I_walked_bwd": 0.0}
    b = wl.base
    with torch.no_grad():
        for rast in wl.rasts:
            _, radii, _, _, pw, saved = R._backend.forward(rast.raster_settings, R.WODILATE, True, b["means3D"], b["scales"],
                                                           b["rotations"], b["opacities"].reshape(-1), b["colors"])
            offs = R.tile_offsets_of(saved, W, H).to(torch.int64)
            L = (offs[1:] - offs[:-1]).view(gy, gx)
            nc = torch.zeros(gy * 16, gx * 16, dtype=torch.int64, device=dev)
            nc[:H, :W] = saved["n_contrib"]
            fT = torch.zeros(gy * 16, gx * 16, device=dev)
            fT[:H, :W] = saved["final_T"]
            t_nc = nc.view(gy, 16, gx, 16).amax(dim=(1, 3))
            t_open = (fT.view(gy, 16, gx, 16) >= 0.01).any(dim=3).any(dim=1)
            fwd = torch.where(t_open, L, torch.minimum(L, (t_nc + 64) // 64 * 64))
            acc["V_live"] += float((pw > 0).sum())
            acc["I_walked_bwd"] += float(t_nc.sum())
            acc["I_walked_fwd"] += float(fwd.sum())
            del saved
    return {k: v / len(wl.rasts) for k, v in acc.items()}


def one_rank_rccl_leg(args, N):
    """Rank 0's multi-GPU step on THIS box: the same workload through a process group of ONE rank over RCCL, forced through
    every collective (LOGRAST_DIST_SINGLE_RANK=1: view groups, hinted pack, all-to-all, unpack-add on the side stream, the
    closing all-gather -- everything an N > 1 step does except the links; profiles/r06_rccl_one_rank.md).  Run as a child
    process with a time limit, after the headline is measured: whatever happens to it, the line above stands."""
    import tempfile
    out = {"what": "the N > 1 step with its whole gradient exchange, one rank over RCCL (no links): local cost of the exchange"}
    fd, path = tempfile.mkstemp(suffix=".json")
    os.close(fd)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--gaussians", str(N), "--width", str(args.width),
           "--height", str(args.height), "--views", str(args.views), "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
           "--no-dropin-mode", "--no-secondary", "--no-forward-only", "--no-rand-variant", "--no-trained-like",
           "--no-one-rank-leg", "--full-out", path]
    try:
        env = dict(os.environ, LOGRAST_DIST_SINGLE_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
            env.pop(k, None)
        p = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=150)
        if p.returncode != 0:
            out["error"] = "exit %d: %s" % (p.returncode, (p.stderr or "")[-300:])
            return out
        with open(path) as f:
            r = json.load(f)
        ex = r.get("exchange") or {}
        out.update(ms_per_step=r.get("ms_per_step"), value=r.get("value"), exchange_mode=ex.get("mode"), parts=ex.get("parts"),
                   backend=ex.get("backend"), streamed=ex.get("streamed"), timing_ms=ex.get("timing_ms"),
                   exchange_only_ms_per_step=ex.get("exchange_only_ms_per_step"), hint_check=ex.get("hint_check"),
                   row_bounds=ex.get("row_bounds"))
    except Exception as e:                      # noqa: BLE001 -- a secondary leg: report, never raise
        out["error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass
    return out


def dist_on(world):
    """More than one rank -- or LOGRAST_DIST_SINGLE_RANK=1 (diagnostics on a one-GPU box: the step's whole exchange runs
    through a ONE-rank RCCL process group, every collective a real RCCL call on the device; the driver never sets it)."""
    return world > 1 or os.environ.get("LOGRAST_DIST_SINGLE_RANK", "0") == "1"


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
        if getattr(args, "scene", "random") == "trained":
            # the statistics of a trained model instead of check_gui's uniform draws (log_amd.scenes.trained_like_scene)
            self.sc = scenes.trained_like_scene(N, seed=0)
        else:
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
        self.last_radii = None
        self.last_weight = None

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
        self.last_radii = out[1]
        self.last_weight = out[4] if len(out) > 4 else None     # point_weight: zero exactly where the view touched nothing
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
    rank = dist.get_rank() if dist_on(world) else 0
    auto_parts = int(args.exchange_parts) <= 0
    parts = max(1, min(len(wl.rasts) // S if auto_parts else int(args.exchange_parts), len(wl.rasts) // S)) if dist_on(world) else 1
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    # The step's gradient exchange (log_amd.dist.StepExchange): the rank's views in `parts` consecutive groups with a
    # bucket each; group g's reduce-scatter runs on a side stream under the rendering of group g + 1.
    # N > 1: the buckets carry the per-row seen counts (what an owner-computes optimizer step needs, and what the
    # touched-row-block exchange reads); --exchange auto decides ONCE, from the warm-up, whether the block form can pay
    # (it costs a bitmap all-reduce and a small read-back per group of views): dense when most blocks are touched.
    track = dist_on(world)
    block_rows = 4096 if (track and args.exchange != "dense") else 0
    # fused accumulation: the buckets are row-major (one 64-byte row of running sums per Gaussian: the chain rule's
    # read-modify-write of a live Gaussian is one line instead of five pieces in five arrays; the exchange moves one block)
    row_major = fused and not args.planar_bucket
    ex = StepExchange(N, dev, world, rank, parts=parts, block_rows=block_rows, track_seen=track, timing=track,
                      row_major=row_major)
    compact = {"on": args.exchange == "compact" and dist_on(world)}
    sparse = {"on": args.exchange == "sparse" and dist_on(world) and row_major, "kmax": None, "gather": None}

    gathered = {"t": None}     # streamed sparse exchange: the persistent result of the closing all-gather (bucket 0 stays clean)

    def n_parts():
        """Groups of views in use: all of them when the groups' exchanges are smaller than the step's (row-sparse, touched
        blocks), ONE when --exchange-parts is automatic and the dense form runs (every dense group is full-size)."""
        return parts if (sparse["on"] or compact["on"] or not auto_parts) else 1

    def on_any_rank(flag):
        """A per-rank condition as ONE decision of the job (a rank that raises alone leaves the others in their next collective)."""
        if not dist_on(world):
            return bool(flag)
        t = torch.tensor([1.0 if flag else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item() > 0)

    def sparse_bound(part):
        """The all-to-all's rows per (sender, owner) pair for group `part`: its own list length of the exact exchange + 10 %
        (padded segments travel whole: one bound for all groups -- the longest view's -- padded the others by ~20 %)."""
        per = sparse.get("kmax_part")
        return per[part] if per and part < len(per) else sparse["kmax"]

    def gather(total):
        if sparse["on"] and n_parts() > 1:
            if gathered["t"] is None:
                gathered["t"] = torch.empty(world * ex.buckets[0].Pr, 16, device=dev)
            return ex.all_gather_grads(total, sparse_kmax=sparse["gather"] or "exact", into=gathered["t"])
        return ex.all_gather_grads(total, sparse_kmax=(sparse["gather"] or "exact") if sparse["on"] else None)

    def exchange():
        """The step's exchange: reduce-scatter (dense / touched blocks / touched rows) per group, then the all-gather."""
        for part in range(n_parts()):
            ex.launch(part, compact=compact["on"], kmax=sparse_bound(part) if sparse["on"] else compact.get("kmax"), sparse=sparse["on"])
        gather(ex.finish())
    lanes = []
    for li in range(S):   # every stream: leaf aliases of the (shared, read-only) attributes + its own flat gradient buckets
        leaves = {k: v.detach().requires_grad_(True) for k, v in wl.base.items()}
        bks = ex.buckets if li == 0 else [GradientBucket(N, dev, world, block_rows=block_rows, track_seen=track,
                                                         row_major=row_major) for _ in range(parts)]
        bks[0].attach(leaves)
        lanes.append((leaves, bks))
    lane_views = [wl.rasts[li::S] for li in range(S)]

    def part_of_view(li, j):
        np_ = n_parts()
        return min(j * np_ // max(len(lane_views[li]), 1), np_ - 1)

    lane_graphs = None
    graph_weight, graph_radii = {}, {}
    state = {"clean": False}   # did the previous step leave every bucket all zero (streamed exchange: pack and clear)?

    def step(do_exchange=True):
        main = torch.cuda.current_stream(dev)
        for st in streams:
            st.wait_stream(main)
        # (streamed sparse exchange, one lane: every bucket was cleared by its own pack -- no zero-fill between steps)
        clean = sparse["on"] and n_parts() > 1 and S == 1 and ex.streamed and do_exchange and state["clean"]
        for li, (_, bks) in enumerate(lanes):
            with torch.cuda.stream(streams[li]):
                if clean:
                    ex.begin_step()
                elif li == 0:
                    ex.zero()            # (lane 0's buckets ARE ex.buckets; also drops the previous step's shards / running sums)
                else:
                    for bk in bks:
                        bk.zero()
        state["clean"] = bool(sparse["on"] and n_parts() > 1 and do_exchange)
        for part in range(n_parts()):
            for li, (leaves, bks) in enumerate(lanes):
                mine = [j for j in range(len(lane_views[li])) if part_of_view(li, j) == part]
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
            if dist_on(world):
                # rows this group of views saw: in this workload every view sees the same rows (radii > 0 for the views'
                # common frustum), recorded once per group from the last forward's radii (kept by one_view)
                streamed = bool(sparse["on"] and n_parts() > 1)
                if wl.last_radii is not None:
                    # (graphs: the radii of this group's last view, which its replay has just rewritten in place; streamed: all
                    # of the step's views are counted in ONE pass when finish() reads the counts)
                    last = [j for j in range(len(lane_views[0])) if part_of_view(0, j) == part][-1:]
                    rad = graph_radii.get((0, last[0])) if (lane_graphs is not None and last) else None
                    ex.seen_bucket(part, streamed).mark_seen(rad if rad is not None else wl.last_radii, defer=streamed)
                if do_exchange and streamed and S == 1 and not args.no_pack_hint:
                    # one view in this group: its point_weight says which rows it touched -- the pack reads 4 bytes per row
                    # of the other 94 % instead of 64 (log_amd.dist.GradientBucket.mark_touched)
                    mine = [j for j in range(len(lane_views[0])) if part_of_view(0, j) == part]
                    if len(mine) == 1:
                        w = graph_weight.get((0, mine[0])) if lane_graphs is not None else wl.last_weight
                        if w is not None:
                            ex.buckets[part].mark_touched(w)
                if do_exchange:
                    ex.launch(part, compact=compact["on"], kmax=sparse_bound(part) if sparse["on"] else compact.get("kmax"),
                              sparse=sparse["on"])
        if dist_on(world) and do_exchange:                   # every rank ends the step with the whole gradient sum
            gather(ex.finish())

    # ---- V and I per view, measured once in exact mode (one 4-byte read-back per view) ----
    R.set_instance_capacity(None)
    stats = []
    for rast in wl.rasts:
        out = wl.one_view(rast, lanes[0][0])
        n_inst, over, max_len, n_rect = R.last_state_info(dev)
        stats.append((int((out[1] > 0).sum().item()), n_inst, max_len, n_rect))
        assert not over
        del out
    res = {"row_major_bucket": row_major,
           "V": float(np.mean([s[0] for s in stats])), "I": float(np.mean([s[1] for s in stats])),
           "I_rect": float(np.mean([s[3] for s in stats])), "bucket_floats": int(ex.buckets[0].flat.numel()),
           "exchange_parts": parts}
    res["exchange_parts_auto"] = auto_parts
    cap = int(max(s[1] for s in stats) * 1.02) + 1024
    if sync_free:
        # from here on: no host sync inside forward(); the longest tile list (it picks the sort's multi-block levels)
        # comes from the same measurement, with the same margin
        R.set_instance_capacity(cap, max_tile_len=int(max(s[2] for s in stats) * 1.02) + 64)
    R.overflow_since_reset(dev)         # clear the status block: from here on every forward is recorded in it
    for _ in range(warmup):
        step()
    if dist_on(world):
        # what the step's exchange has to move: rows with a non-zero gradient (any column) and 4096-row blocks holding one
        torch.cuda.synchronize()
        g = ex.buckets[0]
        nz = torch.zeros(g.Ppad, dtype=torch.bool, device=dev)
        for name, c in g.layout:
            nz |= (g.blocks[name].view(g.Ppad, c) != 0).any(dim=1)
        res["exchange_nonzero_row_fraction"] = float(nz.float().mean())
        if block_rows:
            blk = nz.view(-1, block_rows).any(dim=1).float().mean()
            t = torch.tensor([float(blk)], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            res["exchange_touched_block_fraction"] = float(t.item())
            if args.exchange == "auto":
                compact["on"] = res["exchange_touched_block_fraction"] < GradientBucket.DENSE_ABOVE
            if compact["on"]:
                # One exact compact step (its touched-block collectives sized by a read-back each); from then on they are
                # sized from that step's longest list + 15 % with no read-back (log_amd.dist.TouchedBlocks); a step that
                # outgrew the bound is caught after the timed region.
                step()
                torch.cuda.synchronize()
                longest = max([b.touched.kmax for b in ex.buckets if b.touched is not None] or [g.Pr // block_rows])
                compact["kmax"] = min(g.Pr // block_rows, int(longest * 1.15) + 2)
                res["exchange_block_bound"] = compact["kmax"]
        # rows with a non-zero gradient after a rank's views (the warm-up's last step left the gathered SUM in bucket 0: one
        # more step's rendering gives the local sums back)
        if row_major and args.exchange in ("auto", "sparse") and not compact["on"]:
            step(do_exchange=False)
            t = ex.buckets[0].touched_row_fraction().reshape(1)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            res["exchange_touched_row_fraction"] = float(t.item())
            if args.exchange == "auto":
                before = n_parts()
                sparse["on"] = res["exchange_touched_row_fraction"] < 0.5
                if sparse["on"] and n_parts() != before:
                    # the decision changed the grouping (one group per view now): render into the per-view buckets before the
                    # exact exchange below -- its list lengths become the bounds of every later all-to-all, and the union of a
                    # rank's eight views in ONE bucket made them four times what a view needs (padded segments travel whole)
                    step(do_exchange=False)
            if sparse["on"]:
                # one exact row-sparse exchange (list lengths agreed by a max-reduce + read-back); from then on the
                # all-to-all and the all-gather are sized from its longest lists + 10 % with no read-back.  (Every rank
                # runs the same code on the same decisions: an error here is raised on all of them, and the step falls
                # back to the dense exchange, which the warm-up has already run.)
                try:
                    exchange()
                    torch.cuda.synchronize()
                    sparse["kmax"] = int(max(b.sparse_kmax for b in ex.buckets) * 1.1) + 16
                    if n_parts() > 1:
                        sparse["kmax_part"] = [int(b.sparse_kmax * 1.1) + 16 for b in ex.buckets[:n_parts()]]
                    sparse["gather"] = int(ex.gather_kmax * 1.1) + 16
                    assert not ex.compact_overflowed()
                    res["exchange_row_bounds"] = {"per_pair": sparse["kmax"], "per_pair_by_group": sparse.get("kmax_part"),
                                                  "per_owner_gather": sparse["gather"],
                                                  "rows_per_rank": ex.buckets[0].Pr}
                except Exception as e:
                    if args.exchange == "sparse":
                        raise
                    sparse["on"] = False
                    res["exchange_sparse_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
                    torch.cuda.synchronize()
        res["exchange_mode"] = "sparse" if sparse["on"] else ("compact" if compact["on"] else "dense")
        res["exchange_parts"] = n_parts()
        res["exchange_streamed"] = bool(sparse["on"] and n_parts() > 1)
        del nz
        ex.reset_timing()
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
                    with R.accumulate_grads_into(bks[part_of_view(li, j)].views):
                        g = torch.cuda.CUDAGraph()
                        # thread_local: found in round 6 with a one-rank RCCL group -- under the default (global) mode the
                        # process group's WATCHDOG thread, polling the events of the warm-up's collectives, gets
                        # hipErrorStreamCaptureUnsupported from hipEventQuery while this thread captures, and its
                        # exception aborts the process (a race: it needs a collective the watchdog has not retired yet)
                        with torch.cuda.graph(g, pool=pool, stream=streams[li], capture_error_mode="thread_local"):
                            wl.one_view(rast, leaves)
                    gl.append(g)
                    # (the view's point_weight lives in the graph's pool: held here so that no later capture reuses it --
                    # every replay rewrites it in place; the streamed exchange reads it as the pack's hint)
                    graph_weight[(li, j)] = wl.last_weight
                    graph_radii[(li, j)] = wl.last_radii
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
    if dist_on(world):
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    res["t_enqueued"] = time.perf_counter() - t0
    torch.cuda.synchronize()
    if dist_on(world):
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
    if dist_on(world):
        res["exchange_timing"] = ex.timing_summary(steps)
        # The same collectives with nothing to hide under: link time alone, so that a first run on real xGMI separates what
        # the links cost from what the overlap lost (exposed = ms_per_step - rendering; hidden = exchange_only - exposed).
        assert not on_any_rank(ex.compact_overflowed()), "touched-block / row-sparse exchange outgrew its bound inside the timed region: result invalid"
        xsteps, tsum = 3, 0.0
        for _ in range(xsteps):
            step(do_exchange=False)                     # (the rank's own sums again: the exchange leaves the total in bucket 0)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            tx = time.perf_counter()
            exchange()
            torch.cuda.synchronize()
            tsum += time.perf_counter() - tx
        dist.barrier()
        txe = torch.tensor([tsum], device=dev, dtype=torch.float64)
        dist.all_reduce(txe, op=dist.ReduceOp.MAX)
        res["exchange_only_ms_per_step"] = 1e3 * float(txe.item()) / xsteps
        if sparse["on"] and n_parts() > 1 and S == 1 and not args.no_pack_hint and gathered["t"] is not None:
            # self-check of the hinted pack (the views' point_weight tensors live in the graphs' pools): the step's gathered
            # gradient sum with the hints against the same step with the scanning pack -- the same rows, the same sums up to
            # the order of the reverse walk's float atomics between two renderings
            step()
            torch.cuda.synchronize()
            with_hint = gathered["t"].clone()
            args.no_pack_hint = True
            try:
                step()
                torch.cuda.synchronize()
            finally:
                args.no_pack_hint = False
            scan = gathered["t"]
            rows_h, rows_s = (with_hint != 0).any(1), (scan != 0).any(1)
            # asserted on dL/dopacity and dL/dcolour (row slots 10-13: plain sums of the reverse walk); the slots behind the chain
            # rule are reported only -- on check_gui's uniform scales one needle-shaped Gaussian can carry the tensor's norm and
            # amplify the reverse walk's summation-order noise to 1e-3 between ANY two renderings (seen: 200 000 Gaussians,
            # 1.3e-3 in three of five processes with every row present in both; tests/gpu_util.py quantifies that per row)
            direct = float((with_hint[:, 10:14] - scan[:, 10:14]).norm() / scan[:, 10:14].norm().clamp_min(1e-30))
            res["exchange_hint_check"] = {
                "rel_l2": direct, "rel_l2_all_slots": float((with_hint - scan).norm() / scan.norm().clamp_min(1e-30)),
                "rows_with_hint": int(rows_h.sum()), "rows_scanning": int(rows_s.sum()),
                "rows_differ": int((rows_h != rows_s).sum())}
            # (every rank must take the same decision, or the ones that go on hang in the next collective)
            hc = res["exchange_hint_check"]
            okf = torch.tensor([1.0 if (hc["rel_l2"] < 1e-4 and hc["rows_with_hint"] > 0
                                        and hc["rows_differ"] <= 1e-3 * hc["rows_scanning"]) else 0.0], device=dev)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            res["exchange_hint_check"]["ok_on_every_rank"] = bool(okf.item() > 0)
            assert res["exchange_hint_check"]["ok_on_every_rank"], res["exchange_hint_check"]
            del with_hint, scan
        cols = sum(c for _, c in ex.buckets[0].layout) + 1
        frac = 1.0
        if compact["on"] and ex.touched is not None:
            frac = ex.touched.fraction
        res["exchange_bytes_per_step"] = int(2 * 4 * cols * ex.buckets[0].Ppad * frac * (world - 1) / world)
        if sparse["on"]:
            from log_amd.dist import SPARSE_FLOATS
            b0 = ex.buckets[0]
            res["exchange_bytes_per_step"] = int(4 * (world - 1) * (parts * SPARSE_FLOATS * max(b.sparse_kmax for b in ex.buckets)
                                                                    + b0.Pr + SPARSE_FLOATS * ex.gather_kmax))
    if dist_on(world):
        assert not on_any_rank(ex.compact_overflowed()), "touched-block exchange outgrew its bound inside the timed region: result invalid"
    chk = R.overflow_since_reset(dev)   # every forward since the capacity was set, on all streams
    assert not on_any_rank(chk["overflowed"] or chk["max_instances"] > cap), \
        "tile-instance capacity overflow inside the timed region: result invalid (%r)" % (chk,)
    R.set_instance_capacity(None)
    wl.zero_means2d = True
    if dist_on(world):
        t = torch.tensor([res["elapsed"]], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res["elapsed"] = float(t.item())
    del lane_graphs, lanes, ex
    return res


def mode_summary(N, views, world, steps, r):
    return {"ms_per_view": 1e3 * r["elapsed"] / (steps * views), "ms_per_step": 1e3 * r["elapsed"] / steps,
            "value": float(N) * views * world * steps / r["elapsed"],
            "host_enqueue_ms_per_view": 1e3 * r["t_enqueued"] / (steps * views)}


def R_forms():
    """Which compositing kernels the last forward / backward of this process launched (rows = row-split form)."""
    from log_amd import rasterizer as R
    return R._backend.last_forms or {}


def whole_view_summary(rep, ms_view, best_copy):
    """One view as a whole against the roofs: SURVEY 8d's formula bytes, and the effective bytes (units really processed)."""
    g_formula, g_eff = rep["algorithmic_GBs_whole_view_survey_formula"], rep["algorithmic_GBs_whole_view"]
    return {"ms_per_view": ms_view,
            "survey_formula_bytes": rep["algorithmic_bytes_per_view_survey_formula"], "survey_formula_GBs": g_formula,
            "survey_formula_frac_of_hbm_peak": g_formula / HBM_PEAK_GBS,
            "survey_formula_frac_of_measured_stream_copy": g_formula / best_copy,
            "survey_formula_frac_of_guide_float4_copy": g_formula / GUIDE_COPY_GBS,
            "effective_bytes": rep["algorithmic_bytes_per_view"], "effective_GBs": g_eff,
            "effective_frac_of_measured_stream_copy": g_eff / best_copy,
            "note": "survey_formula counts every Gaussian and every list entry (184 N + 116 V + 108 I + 48 Px); effective "
                    "counts the list entries really walked and the Gaussians really composited"}


def auto_streams(args, N):
    S = args.streams if args.streams > 0 else (3 if N <= 4_000_000 else 1)
    return max(1, min(S, args.views))


def view_bytes(N, V, I, Px, eff=None):
    """Whole-view algorithmic bytes: SURVEY 8d's formula, and the same with the effective units of effective_units()."""
    total = 184 * N + 116 * V + 108 * I + 48 * Px
    if eff is None:
        return total, total
    _, e = algorithmic_bytes(N, V, I, Px, eff)
    return total, sum(e.values())


def workload_report(args, wl, r, N, Px, world, steps, roofs, timing, label):
    """What one measured workload contributes to the line: units, bytes (formula and effective), fractions of the measured
    roofs, per-kernel table and the dominant kernel's roofline object."""
    V, I = r["V"], r["I"]
    eff = effective_units(wl)
    alg, alg_eff = algorithmic_bytes(N, V, I, Px, eff)
    total, total_eff = view_bytes(N, V, I, Px, eff)
    ms_view = 1e3 * r["elapsed"] / (steps * args.views)
    out = {"workload": label, "visible_per_view": V, "tile_instances_per_view": I,
           "tile_instances_per_view_reference_rect_rule": r["I_rect"],
           "effective_units_per_view": dict(eff, note="V_live: point_weight > 0; I_walked_bwd: list entries up to each tile's "
                                                      "deepest contributor; I_walked_fwd: estimated (see bench.py)"),
           "algorithmic_bytes_per_view": total_eff, "algorithmic_bytes_per_view_survey_formula": total,
           "algorithmic_GBs_whole_view": total_eff / (ms_view * 1e-3) / 1e9,
           "algorithmic_GBs_whole_view_survey_formula": total / (ms_view * 1e-3) / 1e9}
    for name, gbs in roofs.items():
        out["algorithmic_frac_of_" + name] = out["algorithmic_GBs_whole_view"] / gbs
        out["algorithmic_frac_of_" + name + "_survey_formula"] = out["algorithmic_GBs_whole_view_survey_formula"] / gbs
    out["algorithmic_frac_of_hbm_peak"] = out["algorithmic_GBs_whole_view"] / HBM_PEAK_GBS
    if timing:
        prof = r["prof_serial"] if r["prof_serial"] is not None else r["prof_timed"]
        out["kernels"], out["roofline"] = roof(prof, alg, alg_eff, N, wl.W, wl.H)
        if r["prof_serial"] is None:
            out["roofline"]["measured"] = "HIP events on the launch stream inside the timed region (one stream)"
        else:
            out["roofline"]["measured"] = ("HIP events on the launch stream, one eagerly launched single-stream step run right "
                                           "after the timed region (inside it the views are replayed HIP graphs and / or "
                                           "several views are in flight)")
            if r["prof_timed"] is not None:
                out["kernels_timed_region"], out["roofline_timed_region"] = roof(r["prof_timed"], alg, alg_eff, N, wl.W, wl.H)
    return out


def forward_only(args, wl, steps=3):
    """torch.no_grad() rendering through the drop-in module, as the reference's demo / validation loop drives it
    (apps/train.py:53-59,100-108 time renderer.vis with CUDA events): ms per view and fps, in the package's default mode
    (speculative stage 2, one read-back per forward) and with a capacity hint (no read-back at all)."""
    import torch
    from log_amd import rasterizer as R
    dev = wl.dev

    def loop(n):
        with torch.no_grad():
            for _ in range(n):
                for rast in wl.rasts:
                    means2D = torch.zeros(wl.N, 3, device=dev)       # renderer.py:135 allocates it for every view
                    rast(means3D=wl.base["means3D"], means2D=means2D, shs=None, colors_precomp=wl.base["colors"],
                         opacities=wl.base["opacities"], scales=wl.base["scales"], rotations=wl.base["rotations"],
                         cov3D_precomp=None)

    def timed(n):
        loop(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(n)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / (n * len(wl.rasts))
        return {"ms_per_view": ms, "fps": 1e3 / ms, "gaussians_per_s": wl.N / (ms * 1e-3)}

    R.set_instance_capacity(None)
    R.overflow_since_reset(dev)
    out = {"default_mode": timed(steps)}
    chk = R.overflow_since_reset(dev)
    R.set_instance_capacity(int(chk["max_instances"] * 1.02) + 1024, max_tile_len=int(chk["max_tile_len"] * 1.02) + 64)
    try:
        out["capacity_hint"] = timed(steps)
        assert not R.overflow_since_reset(dev)["overflowed"]
    finally:
        R.set_instance_capacity(None)
    return out


def c5_band(args, dev, bands=8, band=3, N=100_000_000, W=3840, H=2160, views=2):
    """BASELINE configs[4] on ONE GPU: what one of the 8 ranks does per view when the 4K image is split into bands of tile
    rows (SURVEY 8e second axis) -- the band pre-pass over all N (lograst_tile_rows, identical on every rank), the selection
    of the band's Gaussians, the gather of their attributes, forward + backward of the band, and the scatter-add of the
    band's gradients into the dense per-step bucket rows.  100 M random Gaussians generated on the device (same
    distribution as the headline's: xyz U(-0.5, 0.5), scales U(0, 0.5 N^-1/3), unit quaternions, opacity 0.999)."""
    import numpy as np
    import torch
    from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer
    from log_amd import dist as D, rasterizer as R, scenes
    gen = torch.Generator(device=dev).manual_seed(0)
    smax = 0.5 * float(N) ** (-1.0 / 3.0)
    base = {"means3D": torch.rand(N, 3, device=dev, generator=gen) - 0.5,
            "scales": torch.rand(N, 3, device=dev, generator=gen) * smax,
            "rotations": torch.nn.functional.normalize(torch.rand(N, 4, device=dev, generator=gen) + 1e-3),
            "opacities": torch.full((N, 1), 0.999, device=dev), "colors": torch.rand(N, 3, device=dev, generator=gen)}
    T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
    cams = scenes.orbit_cameras(views, W=W, H=H, focal=2139.0 * W / 1920.0)
    rows = D.band_rows(band, bands, H)
    b, e = D.band_pixels(band, bands, H)
    wloss = torch.rand(3, e - b, W, device=dev, generator=gen)
    sink = {k: torch.zeros_like(v) for k, v in base.items()}                 # the rank's dense per-step gradient rows
    rasts = []
    for cam in cams:
        rs = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam["FoVx"] * 0.5), tanfovy=math.tan(cam["FoVy"] * 0.5),
            bg=T([1.0, 1.0, 1.0]), scale_modifier=1.0, viewmatrix=T(cam["world_view_transform"]),
            projmatrix=T(cam["full_proj_transform"]), sh_degree=0, campos=T(cam["camera_center"]), prefiltered=False,
            debug=False)
        rasts.append(GaussianRasterizer(raster_settings=rs))
    stages, info = {}, {}

    def one(rast, clock):
        clock("band_prepass")
        y0, y1 = rast.tile_rows(base["means3D"], base["scales"], base["rotations"])
        clock("select")
        idx = D.band_index(y0, y1, band, bands, H)
        clock("gather")
        leaves = {k: v[idx].requires_grad_(True) for k, v in base.items()}
        clock("rasterize_fwd_bwd")
        n = int(idx.numel())
        means2D = torch.empty(n, 3, device=dev).requires_grad_(True)
        with R.tile_rows(*rows):
            out = rast(means3D=leaves["means3D"], means2D=means2D, shs=None, colors_precomp=leaves["colors"],
                       opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
                       cov3D_precomp=None)
        out[0][:, b:e].backward(gradient=wloss)
        clock("scatter_grads")
        for k in sink:
            sink[k].index_add_(0, idx, leaves[k].grad)
        clock(None)
        info.update(n_band=n, instances=R.last_state_info(dev)[0])

    def run(staged):
        cur = {"t": None, "name": None}

        def clock(name):
            if staged:
                torch.cuda.synchronize()
                now = time.perf_counter()
                if cur["name"] is not None:
                    stages[cur["name"]] = stages.get(cur["name"], 0.0) + (now - cur["t"])
                cur["t"], cur["name"] = now, name
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rast in rasts:
            one(rast, clock)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / len(rasts)

    run(False)                                   # warm-up: allocator, capacity history of this resolution and band
    ms = run(False)
    stages.clear()
    run(True)
    # the same band rendered from ALL Gaussians (what round 2 did: the rect is clipped inside the full projection)
    def full(rast):
        leaves = {k: v.detach().requires_grad_(True) for k, v in base.items()}
        means2D = torch.empty(N, 3, device=dev).requires_grad_(True)
        with R.tile_rows(*rows):
            out = rast(means3D=leaves["means3D"], means2D=means2D, shs=None, colors_precomp=leaves["colors"],
                       opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
                       cov3D_precomp=None)
        out[0][:, b:e].backward(gradient=wloss)
    def timed(fn):
        fn(rasts[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rast in rasts:
            fn(rast)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / len(rasts)

    ms_full = timed(full)
    # ... and with the band's gradients added straight into the rank's dense per-step rows (what a rank of the node does:
    # no per-view gradient tensors of 100 M rows, no pre-pass, no gather / scatter -- the projection itself drops the
    # Gaussians whose rect misses the band, at 44 bytes each: lr_project_batched_kernel<SPARSE>)

    row_sink = {"rows": torch.zeros(N, 16, device=dev)}           # (row-major: one 64-byte row of running sums per Gaussian)

    def full_sink(rast):
        leaves = {k: v.detach().requires_grad_(True) for k, v in base.items()}
        means2D = torch.empty(N, 3, device=dev).requires_grad_(True)
        with R.accumulate_grads_into(row_sink), R.tile_rows(*rows):
            out = rast(means3D=leaves["means3D"], means2D=means2D, shs=None, colors_precomp=leaves["colors"],
                       opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
                       cov3D_precomp=None)
            out[0][:, b:e].backward(gradient=wloss)

    ms_sink = timed(full_sink)
    from log_amd import tune
    from log_amd import _lib

    def kernels():
        _lib.profile_reset()
        _lib.profile_enable(True)
        full_sink(rasts[0])
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        return {k: round(1e3 * ms_k / max(cnt, 1), 1) for k, (ms_k, cnt) in _lib.profile_read().items()}

    kern = kernels()
    sparse_knob = tune.get_knob("LOGRAST_BAND_SPARSE")
    tune.set_knob("LOGRAST_BAND_SPARSE", 0)
    try:
        ms_sink_dense = timed(full_sink)
        kern_dense = kernels()
    finally:
        tune.set_knob("LOGRAST_BAND_SPARSE", sparse_knob)
    return {"workload": "C5 band (BASELINE.json configs[4] on one of its 8 GPUs): %d random Gaussians (device RNG, seed 0, "
                        "opacity 0.999), %dx%d, band %d of %d = tile rows [%d, %d), %d orbit views: tile-row pre-pass over "
                        "all Gaussians -> select -> gather -> forward + backward of the band -> scatter-add of the gradients"
                        % (N, W, H, band, bands, rows[0], rows[1], views),
            "gaussians": N, "gaussians_in_band": info["n_band"], "tile_instances_in_band": info["instances"],
            "ms_per_view": ms, "stages_ms": {k: 1e3 * v / len(rasts) for k, v in stages.items()},
            "ms_per_view_band_clipped_inside_full_projection": ms_full,
            "ms_per_view_band_clipped_gradient_sink": ms_sink,
            "ms_per_view_band_clipped_gradient_sink_full_view_kernel": ms_sink_dense,
            "kernels_us_band_clipped_gradient_sink": kern,
            "kernels_us_band_clipped_gradient_sink_full_view_kernel": kern_dense,
            "gaussians_per_s_per_gpu": N / (min(ms, ms_full, ms_sink) * 1e-3),
            "note": "ms_per_view: pre-pass + gather / scatter path; band_clipped_*: all 100 M Gaussians handed to the "
                    "rasterizer with tile_row_begin/end set -- autograd gradients (five fresh 100 M-row tensors per view) "
                    "or added into the rank's per-step rows (gradient_sink: the multi-GPU step's form); "
                    "gaussians_per_s_per_gpu is the fastest of the three"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args))
    keep_stdout_for_the_line()
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
    if dist_on(world):
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1 and "MASTER_PORT" not in os.environ:     # (single-rank diagnostics without a launcher)
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    n_ranks = dist.get_world_size() if dist_on(world) else 1     # what the process group really holds

    N, W, H = args.gaussians, args.width, args.height
    Px = W * H
    timing = not args.no_kernel_timing
    fused = not args.no_fused_accumulate
    wl = RasterWorkload(args, N, dev, rank, world, torch, np)
    S = auto_streams(args, N)
    r = measure(args, wl, S, fused, args.steps, args.warmup, world, timing, graphs=not args.no_graphs)
    V, I, I_rect, elapsed = r["V"], r["I"], r["I_rect"], r["elapsed"]
    head = mode_summary(N, args.views, world, args.steps, r)
    op_name = "rand" if args.opacity < 0 else args.opacity

    def label(n, op):
        return ("%s: %d random Gaussians (seed 0, opacity %s), %dx%d, %d orbit views per GPU, fwd+bwd of sum(image*w), "
                "wodilate (5-tuple) flavour" %
                ("north-star point (BASELINE.json north_star / configs[3] per GPU)" if n >= 30_000_000 else
                 ("C2 (configs[1])" if n == 1_000_000 else "custom"), n, op, W, H, args.views))

    result = {
        "metric": "Gaussians/sec fwd+bwd @1080p", "value": head["value"], "unit": "Gaussians/s", "n_gpus": n_ranks,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": label(N, op_name),
            "gaussians": N, "width": W, "height": H, "views_per_gpu": args.views, "opacity": str(op_name), "scene": args.scene,
            "visible_per_view": V, "tile_instances_per_view": I, "tile_instances_per_view_reference_rect_rule": I_rect,
            "streams_per_gpu": S, "fused_gradient_accumulation": fused, "row_major_gradient_bucket": r.get("row_major_bucket"),
            "parallelism": ("view-sharded dp%d, %s reduce-scatter of %d floats/step in %d groups of views (each under the "
                            "next group's rendering, side stream) + one all-gather" %
                            (n_ranks, r.get("exchange_mode", "dense"), r["bucket_floats"], r["exchange_parts"])
                            if dist_on(world) else "single GPU") + ", %d view(s) in flight per GPU (HIP streams)" % S,
        },
        "ms_per_view": head["ms_per_view"], "host_enqueue_ms_per_view": head["host_enqueue_ms_per_view"],
        "modes": {"pipelined": dict(head, streams=S, sync_free=True, fused_gradient_accumulation=fused, hip_graphs=r["graphs"],
                                    note="value of this line: capacity from the warm-up, no host sync in forward(), "
                                         "gradients added by the backward kernels into the step's bucket")},
    }
    if dist_on(world):
        result["exchange"] = {
            "mode": r.get("exchange_mode"), "policy": args.exchange, "parts": r["exchange_parts"], "backend": backend,
            "streamed": r.get("exchange_streamed"), "parts_policy": "auto" if r.get("exchange_parts_auto") else "fixed",
            "nonzero_gradient_row_fraction": r.get("exchange_nonzero_row_fraction"),
            "touched_row_fraction": r.get("exchange_touched_row_fraction"), "row_bounds": r.get("exchange_row_bounds"),
            "sparse_error": r.get("exchange_sparse_error"),
            "touched_4096_row_block_fraction": r.get("exchange_touched_block_fraction"),
            "bytes_moved_per_rank_per_step": r.get("exchange_bytes_per_step"), "timing_ms": r.get("exchange_timing"),
            "exchange_only_ms_per_step": r.get("exchange_only_ms_per_step"),
            "hint_check": r.get("exchange_hint_check"),
            "exchange_only_busbw_GBs": (r["exchange_bytes_per_step"] / (r["exchange_only_ms_per_step"] * 1e-3) / 1e9
                                        if r.get("exchange_only_ms_per_step") and r.get("exchange_bytes_per_step") else None),
            "rccl_env": {k: os.environ.get(k) for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "NCCL_ALGO",
                                                         "NCCL_PROTO", "RCCL_MSCCL_ENABLE") if os.environ.get(k)}}
    if not args.no_dropin_mode:
        # the drop-in default: what LoG's unmodified renderer.py gets -- one stream, the package's default forward (stage 2
        # enqueued speculatively, one read-back per forward that the stream does not wait for), a zero-filled means2D per
        # view, gradients added in place into the leaves' existing .grad
        dsteps = max(1, min(args.steps, 3))
        rd = measure(args, wl, 1, False, dsteps, 1, world, False, sync_free=False)
        from log_amd import rasterizer as R
        result["modes"]["dropin_default"] = dict(
            mode_summary(N, args.views, world, dsteps, rd), streams=1, sync_free=False,
            fused_gradient_accumulation=False, steps=dsteps, speculative_forward=True,
            capacity_retries=R.capacity_stats(reset=False),
            note="one stream, default forward of the package (speculative stage 2 + one read-back per forward on a side "
                 "stream), gradients added in place into the leaves' .grad")

    if rank == 0:
        best_copy, copy_forms = measured_stream_copy_bandwidth(dev)
        roofs = {"measured_stream_copy": best_copy, "measured_copy": measured_copy_bandwidth(dev)}
        result["measured_stream_copy_GBs"] = roofs["measured_stream_copy"]
        result["measured_copy_GBs"] = roofs["measured_copy"]
        rep = workload_report(args, wl, r, N, Px, world, args.steps, roofs, timing, label(N, op_name))
        if "roofline" in rep:
            # Everything a reader needs to redo the fractions sits INSIDE the roofline object (the driver's record keeps
            # this object whole): both denominators, the whole view next to the dominant kernel, A0, and -- filled in
            # below -- the opacity = rand variant.
            rf = rep["roofline"]
            rf["measured_stream_copy_GBs"] = best_copy
            rf["stream_copy_forms_GBs"] = copy_forms
            rf["guide_float4_copy_GBs"] = GUIDE_COPY_GBS
            rf["torch_copy_GBs"] = roofs["measured_copy"]
            rf["frac_of_measured_stream_copy"] = rf["achieved"] / best_copy
            rf["frac_of_guide_float4_copy"] = rf["achieved"] / GUIDE_COPY_GBS
            rf["traffic_frac_of_measured_stream_copy"] = (rf["traffic"] / (rf["avg_launch_us"] * 1e-6) / 1e9 / best_copy
                                                          if rf.get("traffic") else None)
            rf["whole_view"] = whole_view_summary(rep, head["ms_per_view"], best_copy)
            rf["compute_radius"] = compute_radius_leg(wl)
            rf["forms"] = dict(R_forms())
        result["config"]["ms_per_view"] = head["ms_per_view"]
        for k in ("effective_units_per_view", "algorithmic_bytes_per_view", "algorithmic_bytes_per_view_survey_formula",
                  "algorithmic_GBs_whole_view", "algorithmic_GBs_whole_view_survey_formula",
                  "algorithmic_frac_of_measured_stream_copy", "algorithmic_frac_of_measured_stream_copy_survey_formula",
                  "algorithmic_frac_of_measured_copy", "algorithmic_frac_of_measured_copy_survey_formula",
                  "algorithmic_frac_of_hbm_peak", "kernels", "roofline", "kernels_timed_region", "roofline_timed_region"):
            if k in rep:
                result[k] = rep[k]
        if r.get("graphs_error"):
            result["graphs_error"] = r["graphs_error"]
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only: the host cores are shared by all ranks
            result["cpu_baseline"] = cpu_baseline(wl.sc, wl.cams, wl.wloss.cpu().numpy(), N)
    else:
        roofs = {}
    if world == 1 and not args.no_forward_only:
        result["forward_only"] = {"headline": forward_only(args, wl)}
    del wl
    torch.cuda.empty_cache()

    if world == 1 and not args.no_rand_variant and args.opacity >= 0:
        # "opacity = 0.999 AND a second run with opacity = rand(N, 1)" -- same harness, same mode
        try:
            args_r = argparse.Namespace(**dict(vars(args), opacity=-1.0))
            wl_r = RasterWorkload(args_r, N, dev, rank, world, torch, np)
            rsteps = max(2, min(args.steps, 5))
            rr = measure(args_r, wl_r, S, fused, rsteps, max(1, min(args.warmup, 2)), world, timing, graphs=not args.no_graphs)
            mr = dict(mode_summary(N, args.views, world, rsteps, rr), streams=S, sync_free=True, hip_graphs=rr["graphs"],
                      steps=rsteps)
            mr.update(workload_report(args_r, wl_r, rr, N, Px, world, rsteps, roofs, timing, label(N, "rand")))
            result["modes"]["pipelined_opacity_rand"] = mr
            result["config"]["ms_per_view_opacity_rand"] = mr["ms_per_view"]
            if "roofline" in result and "roofline" in mr:
                result["roofline"]["whole_view_opacity_rand"] = whole_view_summary(mr, mr["ms_per_view"], roofs["measured_stream_copy"])
                result["roofline"]["opacity_rand_dominant_kernel"] = {
                    k: mr["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "algorithmic_bytes_per_launch")}
            if not args.no_forward_only:
                result["forward_only"]["headline_opacity_rand"] = forward_only(args_r, wl_r)
            del wl_r
        except Exception as e:
            result["modes"]["pipelined_opacity_rand"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()

    if world == 1 and not args.no_trained_like and args.scene == "random":
        # Round-4 verdict, missing #4: the same 30 M / 1080p / 8-view step on a scene with the statistics of a TRAINED model
        # (log-normal scales sigma 0.5, anisotropy <= 10, sigmoid-normal opacity; tests/test_gpu_scale.py checks the very
        # same scene against the oracle) -- neither an opaque cube where 93 % of the Gaussians never composite nor uniform
        # needles and pancakes.
        try:
            args_t = argparse.Namespace(**dict(vars(args), scene="trained"))
            wl_t = RasterWorkload(args_t, N, dev, rank, world, torch, np)
            tsteps = max(2, min(args.steps, 5))
            rt = measure(args_t, wl_t, S, fused, tsteps, max(1, min(args.warmup, 2)), world, timing, graphs=not args.no_graphs)
            mt = dict(mode_summary(N, args.views, world, tsteps, rt), streams=S, sync_free=True, hip_graphs=rt["graphs"],
                      steps=tsteps)
            mt.update(workload_report(args_t, wl_t, rt, N, Px, world, tsteps, roofs, timing,
                                      "trained-like scene: %d Gaussians (log_amd.scenes.trained_like_scene, seed 0), %dx%d, "
                                      "%d orbit views, same harness as the headline" % (N, W, H, args.views)))
            mt["forms"] = dict(R_forms())
            result["modes"]["pipelined_trained_like"] = mt
            result["config"]["ms_per_view_trained_like"] = mt["ms_per_view"]
            if "roofline" in result:
                result["roofline"]["whole_view_trained_like"] = whole_view_summary(mt, mt["ms_per_view"], roofs["measured_stream_copy"])
            del wl_t
        except Exception as e:
            result["modes"]["pipelined_trained_like"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()

    if world == 1 and not args.no_secondary:
        sec = {}
        try:
            if N != 1_000_000:
                wl2 = RasterWorkload(args, 1_000_000, dev, rank, world, torch, np)
                S2, steps2 = auto_streams(args, 1_000_000), max(args.steps, 20)   # (a 0.65 ms view: 20 steps = 0.1 s)
                r2 = measure(args, wl2, S2, fused, steps2, max(args.warmup, 2), world, timing, graphs=not args.no_graphs)
                c2 = {"modes": {"pipelined": dict(mode_summary(1_000_000, args.views, 1, steps2, r2), streams=S2,
                                                  hip_graphs=r2["graphs"])}}
                c2.update(workload_report(args, wl2, r2, 1_000_000, Px, 1, steps2, roofs, timing,
                                          "C2 (BASELINE.json configs[1]): 1000000 random Gaussians (seed 0, opacity %s), "
                                          "%dx%d, %d orbit views, same harness as the headline" % (op_name, W, H, args.views)))
                if not args.no_dropin_mode:
                    rd2 = measure(args, wl2, 1, False, 3, 1, world, False, sync_free=False)
                    c2["modes"]["dropin_default"] = dict(mode_summary(1_000_000, args.views, 1, 3, rd2), streams=1)
                if not args.no_forward_only:
                    c2["forward_only"] = forward_only(args, wl2, steps=10)
                sec["c2"] = c2
                del wl2
                torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_log_step
            sec["c3"] = bench_log_step.c3_pipeline(views=4, with_torch=args.c3_torch, dev=dev,
                                                   forward_only=not args.no_forward_only)
            torch.cuda.empty_cache()
            if not args.no_c5_band:
                sec["c5_band"] = c5_band(args, dev)
        except Exception as e:   # the headline stands on its own; say what happened to the rest
            sec["error"] = "%s: %s" % (type(e).__name__, e)
        result["secondary"] = sec

    if world == 1 and not dist_on(world) and not args.no_one_rank_leg and not args.no_secondary and N >= 1_000_000:
        result["multi_gpu_step_one_rank_rccl"] = one_rank_rccl_leg(args, N)

    # whatever the libraries left in C stdio buffers (librccl's banner) goes out NOW, on every rank, in front of the line
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if dist_on(world):
        dist.barrier()
    if rank == 0:
        result["parity"] = parity_summary()
        flatten_for_the_driver(result)
        emit(result, args)
    if dist_on(world):
        dist.barrier()
        dist.destroy_process_group()


# ---- what leaves the process -------------------------------------------------------------------------------------------
# Round 5's single line had grown to 26 KB and the driver stopped parsing it (BENCH_r05.json: parsed = null).  Now: the
# FULL result (modes, per-kernel tables, secondary legs) goes to bench_full.json next to this file (and to gpurun_out/
# when that directory exists), and the LAST stdout line is a compact object of the contract keys only, < 4 KB, every key
# <= 40 characters and every string <= 120 (the driver's record truncates beyond that).  tests/test_bench_line_cpu.py
# builds the line from a committed full result and checks size, keys and the JSON round trip.
LINE_LIMIT = 4096
MODE_NOTE = ("value = pipelined step (HIP graphs, capacity hints, gradient sink); "
             "config.dropin_default_ms_per_view = unmodified renderer.py")


def _num(x, digits=6):
    """Floats with `digits` significant digits (the line is for reading; the full file keeps every bit)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float("%.*g" % (digits, float(x)))
    except (TypeError, ValueError):
        return None


def compact_line(result):
    """The contract line from the full result: metric / value / unit / n_gpus / steps / warmup / ms_per_step /
    higher_is_better / scaling / vs_baseline / dtype / data / config{workload + scalars} / roofline / cpu_baseline."""
    g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    cfg, rf, cb = result.get("config") or {}, result.get("roofline"), result.get("cpu_baseline")
    line = {"metric": result["metric"], "value": _num(result["value"], 8), "unit": result["unit"], "mode": MODE_NOTE}
    for k in ("n_gpus", "steps", "warmup"):
        line[k] = result[k]
    line["ms_per_step"] = _num(result["ms_per_step"], 7)
    for k in ("higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        line[k] = result[k]
    short = "%dM %s Gaussians (seed 0, opacity %s), %dx%d, %d orbit views/GPU, fwd+bwd%s" % (
        cfg.get("gaussians", 0) // 1_000_000, "trained-like" if cfg.get("scene") == "trained" else "random",
        cfg.get("opacity", "0.999"), cfg.get("width", 0), cfg.get("height", 0), cfg.get("views_per_gpu", 0),
        "; north_star / configs[3] per GPU" if cfg.get("gaussians", 0) >= 30_000_000 else "")
    c = {"workload": (short if cfg.get("gaussians", 0) >= 1_000_000 else str(cfg.get("workload", "")))[:120]}
    for k in ("gaussians", "width", "height", "views_per_gpu", "streams_per_gpu"):
        c[k] = cfg.get(k)
    c["visible_per_view"] = _num(cfg.get("visible_per_view"), 8)
    c["tile_instances_per_view"] = _num(cfg.get("tile_instances_per_view"), 8)
    c["ms_per_view"] = _num(g(result, "modes", "pipelined", "ms_per_view"))
    c["dropin_default_ms_per_view"] = _num(g(result, "modes", "dropin_default", "ms_per_view"))
    c["ms_per_view_opacity_rand"] = _num(cfg.get("ms_per_view_opacity_rand"))
    c["ms_per_view_trained_like"] = _num(cfg.get("ms_per_view_trained_like"))
    c["forward_only_ms_per_view"] = _num(g(result, "forward_only", "headline", "capacity_hint", "ms_per_view"))
    for k in ("c2_ms_per_view", "c3_ms_per_view", "c3_fused_step_ms_per_view", "c5_band_ms_per_view_gradient_sink"):
        c[k] = _num(cfg.get(k))
    one = result.get("multi_gpu_step_one_rank_rccl")
    if isinstance(one, dict):
        c["rccl_one_rank_step_ms"] = _num(one.get("ms_per_step"))
        c["rccl_one_rank_exchange"] = ("%s x%s" % (one.get("exchange_mode"), one.get("parts"))) if one.get("exchange_mode") else \
            str(one.get("error", ""))[:100]
    c["parallelism"] = str(cfg.get("parallelism", ""))[:120]
    ex = result.get("exchange")
    if isinstance(ex, dict):
        c["exchange_mode"] = ex.get("mode")
        c["exchange_parts"] = ex.get("parts")
        c["exchange_only_ms_per_step"] = _num(ex.get("exchange_only_ms_per_step"))
        c["exchange_bytes_per_rank_per_step"] = ex.get("bytes_moved_per_rank_per_step")
    line["config"] = {k: v for k, v in c.items() if v is not None}
    if isinstance(rf, dict):
        r = {"bound": rf.get("bound"), "kernel": rf.get("kernel"), "achieved": _num(rf.get("achieved")),
             "peak": rf.get("peak"), "unit": rf.get("unit"), "frac": _num(rf.get("frac"), 4),
             "traffic": _num(rf.get("traffic"), 8),
             "traffic_source": (next((w for w in str(rf.get("traffic_source", "")).split() if w.startswith("profiles/")), None)
                                if rf.get("traffic") else None),
             "valu_issue_frac": _num(rf.get("valu_issue_frac"), 3), "avg_launch_us": _num(rf.get("avg_launch_us"), 5),
             "algorithmic_bytes_per_launch": _num(rf.get("algorithmic_bytes_per_launch"), 8),
             "measured_stream_copy_GBs": _num(rf.get("measured_stream_copy_GBs"), 5),
             "frac_of_measured_stream_copy": _num(rf.get("frac_of_measured_stream_copy"), 4),
             "whole_view_frac_formula": _num(rf.get("whole_view_frac_of_measured_stream_copy"), 4),
             "whole_view_frac_effective": _num(rf.get("whole_view_effective_frac_of_measured_stream_copy"), 4),
             "whole_view_rand_frac_formula": _num(rf.get("whole_view_rand_frac_of_measured_stream_copy"), 4),
             "whole_view_trained_frac_formula": _num(rf.get("whole_view_trained_like_frac_of_measured_stream_copy"), 4),
             "whole_view_frac_note": "of measured stream copy; formula = SURVEY 8d bytes, effective = units really processed",
             "compute_radius_us": _num(rf.get("compute_radius_us"), 4),
             "compute_radius_frac": _num(rf.get("compute_radius_frac"), 3),
             "fwd_form": rf.get("fwd_form"), "bwd_form": rf.get("bwd_form")}
        kus = {k[len("kernel_us_"):]: _num(v, 4) for k, v in rf.items() if k.startswith("kernel_us_")}
        for k, v in sorted(kus.items()):
            r["us_" + k[:36]] = v
        line["roofline"] = {k: v for k, v in r.items() if v is not None or k == "traffic"}   # (traffic: a number or null, never absent)
    if isinstance(cb, dict):
        b = {"value": _num(cb.get("value"), 6), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
             "sample": str(cb.get("sample", ""))[:120]}
        for k in ("reference_python_radius_gaussians_per_s", "reference_python_radius_cores",
                  "reference_python_radius_source"):
            if cb.get(k) is not None:
                b[k] = _num(cb[k]) if not isinstance(cb[k], str) else cb[k][:120]
        line["cpu_baseline"] = b
    if parity_line(result.get("parity")) is not None:
        line["parity"] = parity_line(result.get("parity"))
    line["details"] = "bench_full.json (profiles/r06_bench_full.json: the builder's copy of one run)"
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:                    # never let the line outgrow the driver again: shed the per-kernel times
        line["roofline"] = {k: v for k, v in line.get("roofline", {}).items() if not k.startswith("us_")}
    return line


def emit(result, args=None):
    """Full result -> bench_full.json (+ gpurun_out/); compact contract line -> the LAST stdout line."""
    full = json.dumps(result)
    where = ([args.full_out] if args is not None and getattr(args, "full_out", None) else
             [os.path.join(d, "bench_full.json") for d in (ROOT, os.path.join(ROOT, "gpurun_out")) if os.path.isdir(d)])
    for path in where:
        try:
            with open(path, "w") as f:
                f.write(full + "\n")
        except OSError:
            pass
    if args is not None and getattr(args, "print_full", False):
        _to_stdout(full)
    text = json.dumps(compact_line(result), separators=(",", ":"))
    assert len(text) < LINE_LIMIT and "\n" not in text, len(text)
    _to_stdout(text)


_STDOUT_FD = None


def keep_stdout_for_the_line():
    """Everything but the result line goes to stderr -- at the file-descriptor level, for every rank, before any library
    is loaded.  Found in round 6 with a one-rank RCCL group on the MI355X: librccl prints a five-line banner ("RCCL version
    ... Librccl path : ...") through C stdio to STDOUT, which reaches the pipe when the process exits, i.e. AFTER the JSON
    line -- once per rank -- and a reader that takes the last stdout line as the result finds "Librccl path" there."""
    global _STDOUT_FD
    if _STDOUT_FD is None:
        sys.stdout.flush()
        _STDOUT_FD = os.dup(1)
        os.dup2(2, 1)


def _to_stdout(text):
    sys.stdout.flush()
    if _STDOUT_FD is None:
        print(text, flush=True)
        return
    data = (text + "\n").encode()
    while data:
        data = data[os.write(_STDOUT_FD, data):]


def flatten_for_the_driver(result):
    """The driver's record keeps the SCALAR keys of `roofline` and `config` (nested objects are dropped from its `parsed`
    copy): everything the nested objects say that a reader of BENCH_rNN.json needs is repeated as scalars -- the A0 leg,
    the whole view against the measured copy (opaque / opacity = rand / trained-like), which compositing form ran, the
    drop-in default mode, and the secondary legs' per-view times (round-4 verdict, missing #5 / next #8)."""
    rf = result.get("roofline")
    if not isinstance(rf, dict):
        return
    g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    put = lambda k, v: rf.__setitem__(k, v) if v is not None else None
    put("compute_radius_us", g(rf, "compute_radius", "us"))
    put("compute_radius_frac", g(rf, "compute_radius", "frac_of_hbm_peak"))
    put("compute_radius_GBs", g(rf, "compute_radius", "GBs"))
    put("whole_view_ms", g(rf, "whole_view", "ms_per_view"))
    put("whole_view_frac_of_measured_stream_copy", g(rf, "whole_view", "survey_formula_frac_of_measured_stream_copy"))
    put("whole_view_frac_of_hbm_peak", g(rf, "whole_view", "survey_formula_frac_of_hbm_peak"))
    put("whole_view_effective_frac_of_measured_stream_copy", g(rf, "whole_view", "effective_frac_of_measured_stream_copy"))
    put("whole_view_rand_ms", g(rf, "whole_view_opacity_rand", "ms_per_view"))
    put("whole_view_rand_frac_of_measured_stream_copy",
        g(rf, "whole_view_opacity_rand", "survey_formula_frac_of_measured_stream_copy"))
    put("whole_view_rand_effective_frac_of_measured_stream_copy",
        g(rf, "whole_view_opacity_rand", "effective_frac_of_measured_stream_copy"))
    put("whole_view_trained_like_ms", g(rf, "whole_view_trained_like", "ms_per_view"))
    put("whole_view_trained_like_frac_of_measured_stream_copy",
        g(rf, "whole_view_trained_like", "survey_formula_frac_of_measured_stream_copy"))
    put("whole_view_trained_like_effective_frac_of_measured_stream_copy",
        g(rf, "whole_view_trained_like", "effective_frac_of_measured_stream_copy"))
    put("fwd_form", g(rf, "forms", "fwd"))
    put("bwd_form", g(rf, "forms", "bwd"))
    put("trained_like_fwd_form", g(result, "modes", "pipelined_trained_like", "forms", "fwd"))
    put("trained_like_bwd_form", g(result, "modes", "pipelined_trained_like", "forms", "bwd"))
    put("dropin_default_ms_per_view", g(result, "modes", "dropin_default", "ms_per_view"))
    put("forward_only_ms_per_view", g(result, "forward_only", "headline", "capacity_hint", "ms_per_view"))
    for name, k in (result.get("kernels") or {}).items():          # the per-kernel table of the headline, in microseconds
        put("kernel_us_" + name, k.get("avg_us"))
    cfg = result.get("config", {})
    sec = result.get("secondary") or {}
    for key, path in (("c2_ms_per_view", ("c2", "modes", "pipelined", "ms_per_view")),
                      ("c2_dropin_default_ms_per_view", ("c2", "modes", "dropin_default", "ms_per_view")),
                      ("c3_ms_per_view", ("c3", "ms_per_view")),
                      ("c3_fused_step_ms_per_view", ("c3", "ms_per_view_fused_step")),
                      ("c5_band_ms_per_view_gradient_sink", ("c5_band", "ms_per_view_band_clipped_gradient_sink"))):
        v = g(sec, *path)
        if v is not None:
            cfg[key] = v
    for name, k in (g(sec, "c2", "kernels") or {}).items():
        put("c2_kernel_us_" + name, k.get("avg_us"))


def roof(prof, alg, alg_eff, N, W, H):
    """-> (per-kernel table, roofline object of the dominant kernel).  GB/s figures divide the EFFECTIVE algorithmic bytes
    (units the launch really processes) by the average launch duration; the SURVEY-formula bytes are kept beside them."""
    kern = {}
    for name, (ms, cnt) in prof.items():
        kern[name] = {"avg_us": 1e3 * ms / cnt, "launches": int(cnt)}
        if name in alg:   # per-kernel algorithmic GB/s against the same 8 TB/s roof
            t = 1e-3 * ms / cnt
            kern[name].update(alg_GBs=alg_eff[name] / t / 1e9, hbm_frac=alg_eff[name] / t / 1e9 / HBM_PEAK_GBS,
                              alg_bytes=alg_eff[name], alg_bytes_survey_formula=alg[name],
                              hbm_frac_survey_formula=alg[name] / t / 1e9 / HBM_PEAK_GBS)
    sort_ms = sum(prof[k][0] for k in prof if k.startswith("sort"))
    merged = {k: prof[k][0] for k in prof if not k.startswith("sort")}
    if sort_ms:
        merged["sort"] = sort_ms
    dom = max(merged, key=merged.get)
    launches = prof[dom][1] if dom in prof else prof["sort_small"][1]
    avg_s = merged[dom] / launches / 1e3
    achieved = alg_eff.get(dom, 0) / avg_s / 1e9
    return kern, {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                  "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                  "achieved_survey_formula": alg.get(dom, 0) / avg_s / 1e9,
                  "frac_survey_formula": alg.get(dom, 0) / avg_s / 1e9 / HBM_PEAK_GBS,
                  "traffic": pmc_traffic(dom, N, W, H),
                  "traffic_source": ("committed profile %s (collected by the builder with rocprofv3 --pmc on this workload in "
                                     "this round; not measured in this run)" % os.path.relpath(traffic_profile(N, W, H)[0], ROOT)
                                     if traffic_profile(N, W, H)[0] else
                                     "none: no counter profile of this workload from round %d is committed" % BENCH_ROUND),
                  # the compositing kernels are bound by fp32 VALU issue, which the hbm/mfma vocabulary of
                  # this object cannot name: fraction of SIMD issue cycles spent in VALU ops (PMC pass)
                  "valu_issue_frac": pmc_traffic(dom, N, W, H, "valu_active_frac_at_2p4GHz"),
                  "algorithmic_bytes_per_launch": alg_eff.get(dom, 0),
                  "algorithmic_bytes_per_launch_survey_formula": alg.get(dom, 0), "avg_launch_us": avg_s * 1e6}


def measured_copy_bandwidth(dev, mib=1024, reps=5):
    """Device-to-device copy of `mib` MiB with torch's Tensor.copy_ (read + write counted), best of `reps`: GB/s."""
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


GUIDE_COPY_GBS = 6290.0   # the float4 copy /opt/skills/guides/MI355X_MICROARCH.md measured on this chip (79 % of 8 TB/s)
COPY_FORMS = {0: "grid-stride x4, non-temporal", 1: "one access per lane, plain", 2: "grid-stride x8, non-temporal",
              3: "grid-stride x4, plain loads + non-temporal stores", 4: "one access per lane, non-temporal"}


def measured_stream_copy_bandwidth(dev, mib=2048, reps=4):
    """The measured roof: `mib` MiB copied device-to-device by this library's own streaming kernels (lograst_stream_copy:
    16-byte accesses; five forms x a few grid sizes, include/lograst.h), launched on torch's current stream so that the
    torch events bracket them: GB/s with read + write counted.  -> (best GB/s over everything, {form: best GB/s})."""
    import ctypes
    import torch
    from log_amd import _lib
    L = _lib.lib()
    nbytes = mib * 1024 * 1024
    a = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    pa, pb = ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr())
    forms = {}
    for form in COPY_FORMS:
        best = float("inf")
        for blocks in ((0,) if form in (1, 4) else (1024, 2048, 4096, 8192, 16384)):
            arg = (form << 20) | blocks
            _lib.check(L.lograst_stream_copy(pb, pa, nbytes, arg, stream))
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(L.lograst_stream_copy(pb, pa, nbytes, arg, stream))
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1))
        forms[COPY_FORMS[form]] = 2 * nbytes / (best * 1e-3) / 1e9
    del a, b
    return max(forms.values()), forms


def compute_radius_leg(wl, reps=5):
    """A0 (LoG/cuda/compute_radius_kernel.cu:107-183 -> lograst_compute_radius) on the headline's Gaussians: 44 bytes read
    + 4 written per Gaussian, best of `reps`, torch events on the launch stream."""
    import torch
    from log_amd import rasterizer as R
    rs = wl.rasts[0].raster_settings
    fx, fy = rs.image_width / (2.0 * rs.tanfovx), rs.image_height / (2.0 * rs.tanfovy)
    b = wl.base
    call = lambda: R._backend.compute_radius(b["means3D"], b["scales"], b["rotations"], rs.projmatrix, rs.viewmatrix, fx, fy,
                                             rs.tanfovx, rs.tanfovy)
    call()
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = call()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    del out
    nbytes = 48.0 * wl.N
    return {"kernel": "compute_radius (A0)", "gaussians": wl.N, "us": 1e3 * best, "algorithmic_bytes": nbytes,
            "GBs": nbytes / (best * 1e-3) / 1e9, "frac_of_hbm_peak": nbytes / (best * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "gaussians_per_s": wl.N / (best * 1e-3)}


def traffic_profile(N, W, H):
    """-> (path, dict) of the committed counter profile of this exact workload FROM THIS ROUND (profiles/rNN_traffic*.json,
    NN = BENCH_ROUND), or (None, None): a profile of an earlier round describes kernels that have changed since
    (round-5 verdict, #8: `roofline.traffic` must not go stale silently)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r%02d_traffic*.json" % BENCH_ROUND)), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            wl = d.get("workload", {})
            if (wl.get("gaussians"), wl.get("width"), wl.get("height")) == (N, W, H):
                return path, d
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def pmc_traffic(kernel, N, W, H, field="traffic_bytes"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this round (FETCH_SIZE / WRITE_SIZE
    collected and corrected as MI355X_MICROARCH.md prescribes); None when no profile of this exact workload from this
    round is committed -- counters cannot be collected from inside the timed run."""
    _, d = traffic_profile(N, W, H)
    try:
        return d["kernels"][kernel].get(field) if d else None
    except (KeyError, AttributeError):
        return None


def parity_summary():
    """The measured gradient deviations of this round's `pytest -m gpu` run (profiles/rNN_parity_summary.json, written by
    tools/anchor_stats_md.py from the tests' own dumps): HIP against the fp32 oracle and against float64, all rows and
    well-conditioned rows, per case group -- carried in the full result so that what misses north_star's plain 1e-4 (the
    fork's clamp on check_gui's uniform draws) is reported next to the throughput, not buried (round-5 verdict, next #2d)."""
    path = os.path.join(ROOT, "profiles", "r%02d_parity_summary.json" % BENCH_ROUND)
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    d["file"] = os.path.relpath(path, ROOT)
    return d


def parity_line(p):
    """The few numbers of parity_summary() the compact line carries: worst all-rows rel-L2 (HIP vs fp32 oracle) over the C2
    views, per flavour, and on the trained-like 30 M scene."""
    if not isinstance(p, dict):
        return None
    g = p.get("groups", {})
    worst = lambda names: max([g[n][k]["all_rows_hip_vs_oracle"] for n in names if n in g for k in ("means3D", "scales", "rotations")
                               if k in g[n] and "all_rows_hip_vs_oracle" in g[n][k]] or [None], key=lambda x: -1 if x is None else x)
    walk = lambda names: max([g[n][k]["hip_vs_f64"] for n in names if n in g for k in ("means2D", "conic", "opacities", "colors")
                              if k in g[n]] or [None], key=lambda x: -1 if x is None else x)
    out = {"tol": 1e-4, "what": "max rel-L2 over ALL rows, HIP vs fp32 oracle (dL/dmeans3D, dscales, drotations), MI355X",
           "c2_upstream_pkg": _num(worst(("c2_upstream_opaque", "c2_upstream_rand")), 2),
           "c2_wodilate_fork_clamp": _num(worst(("c2_opaque", "c2_rand")), 2),
           "trained_like_30M": _num(worst(("trained_like_30M",)), 2),
           "reverse_walk_vs_f64_c2": _num(walk(("c2_opaque", "c2_rand", "c2_upstream_opaque", "c2_upstream_rand")), 2),
           "file": p.get("file")}
    return {k: v for k, v in out.items() if v is not None}


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
    out = {"value": N * passes / dt, "unit": "Gaussians/s", "cores": cores, "kind": "port",
           "sample": "%d view pass(es) (cycling the %d views), all %d Gaussians each, forward+backward, %.1f s total"
                     % (passes, len(views), N, dt)}
    out.update(reference_python_radius())
    return out


def reference_python_radius():
    """north_star: "alongside the reference's pure-PyTorch/CPU render path timed on the same box's host cores".  The
    reference has no CPU renderer; its only CPU-runnable arithmetic on this path is LoG.model.geometry.compute_radius
    (geometry.py:132-151, the Python twin of A0).  /root/reference does not exist on the driver's box, so the number is the
    builder's measurement on a pool box of the same kind (tools/time_reference_radius.py, which IMPORTS the reference:
    profiles/rNN_reference_python_radius.json), carried here next to the port's whole-path figure."""
    path = os.path.join(ROOT, "profiles", "r%02d_reference_python_radius.json" % BENCH_ROUND)
    try:
        with open(path) as f:
            d = json.load(f)
        return {"reference_python_radius_gaussians_per_s": d["gaussians_per_s"], "reference_python_radius_cores": d["cores"],
                "reference_python_radius_points": d["points"],
                "reference_python_radius_gpu_kernel_gaussians_per_s": d.get("gpu_gaussians_per_s"),
                "reference_python_radius_source": os.path.relpath(path, ROOT) + " (geometry.py:132-151; builder-run on a pool box)"}
    except (OSError, ValueError, KeyError):
        return {}


if __name__ == "__main__":
    main()

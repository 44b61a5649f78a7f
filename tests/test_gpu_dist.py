"""The multi-process path of bench.py on ONE GPU: two ranks sharing the device over gloo (RCCL refuses two ranks on one
device; the driver's runs use RCCL with one GPU per rank).  Exercises the self-spawn of `bench.py --gpus N`, the
per-rank view sharding, the gradient bucket reduction and the n_gpus the line reports."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, env=None):
    """-> the FULL result object (bench.py --print-full: an earlier stdout line), after checking that the LAST stdout line is
    the compact contract object the driver parses (< 4 KB, the same value)."""
    e = dict(os.environ, **(env or {}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--print-full"] + list(flags), env=e,
                         capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) >= 2, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    assert out.stdout.rstrip().splitlines()[-1] == lines[-1] and len(lines[-1]) < 4096, len(lines[-1])
    # nothing but the result on stdout (librccl's banner, warnings, progress: all on stderr -- bench.keep_stdout_for_the_line)
    assert all(l.startswith("{") for l in out.stdout.splitlines() if l.strip()), out.stdout[-600:]
    full, line = json.loads(lines[-2]), json.loads(lines[-1])
    assert line["value"] == pytest.approx(full["value"], rel=1e-6) and line["n_gpus"] == full["n_gpus"]
    full["_line"] = line
    return full


def test_bench_self_spawns_two_ranks_and_reports_them():
    one = _bench("--gpus", "1", "--gaussians", "200000", "--steps", "2", "--warmup", "1", "--no-secondary",
                 "--no-cpu-baseline", "--no-dropin-mode")
    two = _bench("--gpus", "2", "--gaussians", "200000", "--steps", "2", "--warmup", "1", "--no-secondary",
                 "--no-cpu-baseline", "--no-dropin-mode", env={"LOGRAST_DIST_BACKEND": "gloo", "LOGRAST_SHARE_GPU": "1"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["views_per_gpu"] == one["config"]["views_per_gpu"] == 8       # weak scaling: per-GPU work is fixed
    assert "dp2" in two["config"]["parallelism"] and two["value"] > 0
    # both ranks rendered the same Gaussians from their own 8 of the 16 cameras
    assert abs(two["config"]["visible_per_view"] - one["config"]["visible_per_view"]) < 0.05 * one["config"]["visible_per_view"]
    # the exchange: mode decided from the warm-up, the exchange-only leg timed, every form runs without outgrowing its bounds
    assert two["exchange"]["mode"] in ("dense", "sparse") and two["exchange"]["exchange_only_ms_per_step"] > 0
    for mode in ("dense", "sparse"):
        forced = _bench("--gpus", "2", "--gaussians", "200000", "--steps", "2", "--warmup", "1", "--no-secondary",
                        "--no-cpu-baseline", "--no-dropin-mode", "--exchange", mode,
                        env={"LOGRAST_DIST_BACKEND": "gloo", "LOGRAST_SHARE_GPU": "1"})
        assert forced["exchange"]["mode"] == mode and forced["value"] > 0
        if mode == "sparse":
            assert 0 < forced["exchange"]["touched_row_fraction"] <= 1.0
            # round 6: by default the row-sparse exchange is STREAMED, one group per view (a view's touched rows are packed,
            # cleared and sent while the next view renders); --exchange-parts 1 is round 5's one exchange per step
            # (200 000 Gaussians: three views in flight per rank, so 8 // 3 = 2 groups; the 30 M headline: one per view)
            assert forced["exchange"]["streamed"] and forced["exchange"]["parts"] >= 2, forced["exchange"]
            one = _bench("--gpus", "2", "--gaussians", "200000", "--steps", "2", "--warmup", "1", "--no-secondary",
                         "--no-cpu-baseline", "--no-dropin-mode", "--exchange", "sparse", "--exchange-parts", "1",
                         env={"LOGRAST_DIST_BACKEND": "gloo", "LOGRAST_SHARE_GPU": "1"})
            assert not one["exchange"]["streamed"] and one["exchange"]["parts"] == 1 and one["value"] > 0
            for e in (forced["exchange"], one["exchange"]):
                assert e["exchange_only_ms_per_step"] > 0 and e["timing_ms"]["exposed_join_ms_per_step"] >= 0
        else:
            assert forced["exchange"]["parts"] == 1 and not forced["exchange"]["streamed"]


def test_bench_two_ranks_over_rccl_when_the_box_has_two_gpus():
    """The same self-spawn with the default backend (`nccl` = RCCL) and one GPU per rank: runs on the first box that has
    two GPUs without a code change (the pool's boxes expose one: visibly skipped there)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    for mode in ("dense", "sparse"):
        two = _bench("--gpus", "2", "--gaussians", "200000", "--steps", "2", "--warmup", "1", "--no-secondary",
                     "--no-cpu-baseline", "--no-dropin-mode", "--exchange", mode)
        assert two["n_gpus"] == 2 and two["value"] > 0 and two["exchange"]["mode"] == mode
        assert two["exchange"].get("backend", "nccl") == "nccl"


def test_bench_line_carries_the_contract_fields():
    d = _bench("--gaussians", "300000", "--steps", "2", "--warmup", "1", "--no-secondary")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "modes"):
        assert k in d, k
    assert d["dtype"] == "f32" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert set(d["modes"]) == {"pipelined", "dropin_default", "pipelined_opacity_rand", "pipelined_trained_like"}
    assert "error" not in d["modes"]["pipelined_opacity_rand"] and d["modes"]["pipelined_opacity_rand"]["value"] > 0
    assert "error" not in d["modes"]["pipelined_trained_like"] and d["modes"]["pipelined_trained_like"]["value"] > 0
    r = d["roofline"]
    # what the driver's record keeps of the nested objects, as scalars (round-4 verdict, next #8)
    for k in ("compute_radius_us", "compute_radius_frac", "whole_view_frac_of_measured_stream_copy",
              "whole_view_rand_frac_of_measured_stream_copy", "whole_view_trained_like_frac_of_measured_stream_copy",
              "dropin_default_ms_per_view", "kernel_us_project", "kernel_us_blend_bwd"):
        assert isinstance(r[k], float) and r[k] > 0, k
    assert r["fwd_form"] in ("rows", "quadrant") and r["bwd_form"] in ("rows", "quadrant")
    assert d["config"]["ms_per_view_trained_like"] > 0
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # (this test's 300 000-Gaussian workload has no counter profile: traffic is absent and the source says so; the headline's
    # comes from this round's committed profile -- tests/test_bench_line_cpu.py)
    assert r["traffic"] is None and r["traffic_source"].startswith("none")
    # effective algorithmic bytes: no kernel is credited with more than the peak
    for name, k in d["kernels"].items():
        assert k.get("hbm_frac", 0.0) <= 1.0, (name, k)
    assert d["measured_stream_copy_GBs"] > 1000 and d["measured_copy_GBs"] > 1000
    # ... and the compact last line (what BENCH_rNN.json keeps): the contract objects, the mode note up front
    line = d["_line"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "mode"):
        assert k in line, k
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"] or k == "traffic", k
    assert line["config"]["dropin_default_ms_per_view"] > 0 and line["config"]["ms_per_view"] > 0
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
    assert 0 < d["algorithmic_frac_of_measured_stream_copy"] < 1
    assert d["effective_units_per_view"]["I_walked_bwd"] <= d["config"]["tile_instances_per_view"]
    fo = d["forward_only"]["headline"]
    assert fo["default_mode"]["fps"] > 0 and fo["capacity_hint"]["ms_per_view"] > 0
    assert d["modes"]["dropin_default"]["capacity_retries"]["forwards"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1


# ---- log_amd.dist.StepExchange on the device: two ranks (sharing the GPU, gloo) vs one process -----------------------
def _render_views(ex_or_bucket, views, n_views_rank, dev, parts_of=None):
    """Renders `views` (list of (index within the rank, camera)) forward + backward with the gradients added by the
    backward kernels into the bucket of the view's group."""
    import math
    import numpy as np
    import torch
    from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer
    from log_amd import rasterizer as R, scenes
    N, W, H = 60000, 320, 208
    sc = scenes.random_scene(N, seed=5, opacity=None, smax=0.03)
    sc["xyz"] = sc["xyz"].copy()
    sc["xyz"][(np.arange(N) // 256) % 2 == 1] += 1000.0        # every other 256-row block is out of every view
    T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
    leaves = dict(means3D=T(sc["xyz"]), scales=T(sc["scaling"]), rotations=T(sc["rotation"]), opacities=T(sc["opacity"]),
                  colors=T(sc["colors"]))
    leaves = {k: v.requires_grad_(True) for k, v in leaves.items()}
    w = T(np.random.default_rng(2).random((3, H, W), dtype=np.float32))
    for j, cam in views:
        bucket = ex_or_bucket.bucket_of(j, n_views_rank) if hasattr(ex_or_bucket, "bucket_of") else ex_or_bucket
        rs = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam["FoVx"] * 0.5), tanfovy=math.tan(cam["FoVy"] * 0.5),
            bg=T([0, 0, 0]), scale_modifier=1.0, viewmatrix=T(cam["world_view_transform"]),
            projmatrix=T(cam["full_proj_transform"]), sh_degree=0, campos=T(cam["camera_center"]), prefiltered=False,
            debug=False)
        m2 = torch.zeros(N, 3, device=dev, requires_grad=True)
        with R.accumulate_grads_into(bucket.views):
            out = GaussianRasterizer(raster_settings=rs)(
                means3D=leaves["means3D"], means2D=m2, shs=None, colors_precomp=leaves["colors"],
                opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
            out[0].backward(gradient=w)
        bucket.mark_seen(out[1])
        if hasattr(ex_or_bucket, "bucket_of") and j == ex_or_bucket.last_view_of(ex_or_bucket.buckets.index(bucket), n_views_rank):
            ex_or_bucket.launch(ex_or_bucket.buckets.index(bucket), compact=parts_of == "compact")
    return N


def _exchange_worker(rank, world, port, out, mode, backend="gloo"):
    import torch
    import torch.distributed as dist
    from log_amd import scenes
    from log_amd.dist import StepExchange
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if backend == "nccl":
        # ONE rank over RCCL (a pool box has one GPU, and RCCL refuses two ranks on one device): every collective of the
        # exchange is still a real RCCL call on the MI355X -- the branches of log_amd/dist.py that gloo never takes
        os.environ["LOGRAST_DIST_SINGLE_RANK"] = "1"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        cams = scenes.orbit_cameras(8, W=320, H=208, focal=300.0)
        mine = [(j, cams[i]) for j, i in enumerate(range(rank, 8, world))]
        ex = StepExchange(60000, dev, world, rank, parts=2, block_rows=256 if mode == "compact" else 0)
        _render_views(ex, mine, len(mine), dev, parts_of=mode)
        total = ex.finish()
        assert (ex.touched is not None and ex.touched.kmax <= ex.touched.nb // 2 + 1) == (mode == "compact")
        seen_rows = total["seen"].clone()
        flat = ex.all_gather_grads(total).clone()
        torch.cuda.synchronize()
        torch.save({"flat": flat.cpu(), "seen": seen_rows.cpu(), "Pr": ex.buckets[0].Pr}, os.path.join(out, f"x{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["dense", "compact"])
def test_step_exchange_two_ranks_equals_one_process(tmp_path, mode):
    """Eight views: rank r of 2 renders four of them in two groups, every group reduce-scattered from the side stream as
    soon as its last backward is enqueued; after finish() + all_gather_grads both ranks hold the sum one process
    accumulates over all eight views (float atomics: 1e-4 of the largest entry), and the seen counts of their own rows."""
    import socket
    import torch
    import torch.multiprocessing as mp
    from log_amd import scenes
    from log_amd.dist import FlatParams, GradientBucket
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_exchange_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, f"x{r}.pt")) for r in range(world)]
    assert torch.equal(got[0]["flat"], got[1]["flat"])
    dev = torch.device("cuda:0")
    cams = scenes.orbit_cameras(8, W=320, H=208, focal=300.0)
    ref = GradientBucket(60000, dev)
    _render_views(ref, list(enumerate(cams)), 8, dev)
    torch.cuda.synchronize()
    two = GradientBucket(60000, "cpu", world, block_rows=256 if mode == "compact" else 0)
    two.flat.copy_(got[0]["flat"])
    assert float(ref.flat.abs().sum()) > 0
    for name, _ in ref.layout:
        a, b = ref.views[name].cpu(), two.views[name]
        # float atomics in a different order: 1e-4 of the largest entry on the reverse walk's own sums; behind the per-Gaussian
        # chain rule the largest entries belong to ill-conditioned rows that amplify that noise (tests/gpu_util.py: the
        # float64-anchored criteria) -- in L2 over the tensor the two sums still agree to 1e-4
        if name in ("opacities", "colors", "seen"):
            assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-7, name
        else:
            assert float((a - b).abs().max()) <= 1e-3 * float(a.abs().max()) + 1e-7, name
            assert float((a - b).norm()) <= 1e-4 * float(a.norm()) + 1e-7, name
    Pr = got[0]["Pr"]
    seen = ref.seen.cpu()
    for r in range(world):
        rows = seen[r * Pr:(r + 1) * Pr]
        assert torch.equal(got[r]["seen"][:rows.numel()], rows), r


@pytest.mark.parametrize("mode", ["dense", "compact"])
def test_step_exchange_one_rank_over_rccl(tmp_path, mode):
    """The RCCL branches of log_amd/dist.py (reduce_scatter_tensor, all_gather_into_tensor, all_to_all_single, the MAX /
    SUM all-reduces of flags and counts) executed on the device: a process group of ONE rank over `nccl` (= RCCL), forced
    through every collective by LOGRAST_DIST_SINGLE_RANK=1, renders the eight views in two groups and must deliver what the
    plain single-process accumulation delivers (round-5 verdict, missing #3: "the RCCL code path has never executed")."""
    import socket
    import torch
    import torch.multiprocessing as mp
    from log_amd import scenes
    from log_amd.dist import GradientBucket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_exchange_worker, args=(1, port, str(tmp_path), mode, "nccl"), nprocs=1, join=True)
    got = torch.load(os.path.join(tmp_path, "x0.pt"))
    dev = torch.device("cuda:0")
    cams = scenes.orbit_cameras(8, W=320, H=208, focal=300.0)
    ref = GradientBucket(60000, dev)
    _render_views(ref, list(enumerate(cams)), 8, dev)
    torch.cuda.synchronize()
    one = GradientBucket(60000, "cpu", 1, block_rows=256 if mode == "compact" else 0)
    one.flat.copy_(got["flat"])
    assert float(ref.flat.abs().sum()) > 0
    for name, _ in ref.layout:
        a, b = ref.views[name].cpu(), one.views[name][: ref.views[name].shape[0]]
        assert float((a - b).norm()) <= 1e-4 * float(a.norm()) + 1e-7, name
    rows = ref.seen.cpu()
    assert torch.equal(got["seen"][:rows.numel()], rows)


def test_bench_one_rank_over_rccl():
    """bench.py's multi-GPU step (view groups, the exchange on the side stream, the closing all-gather) through a one-rank
    RCCL group, dense and row-sparse (streamed): it runs, reports the exchange, and raises no overflow."""
    for mode in ("dense", "sparse"):
        full = _bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--gaussians", "200000", "--no-cpu-baseline",
                      "--no-dropin-mode", "--no-secondary", "--no-forward-only", "--no-rand-variant", "--no-trained-like",
                      "--exchange", mode, env=dict(LOGRAST_DIST_SINGLE_RANK="1"))
        assert full["n_gpus"] == 1 and full["exchange"]["backend"] == "nccl", full["exchange"]
        assert full["exchange"]["mode"] == mode and full["exchange"]["exchange_only_ms_per_step"] > 0
        assert full["value"] > 0
    # one view in flight (as at the 30 M headline): every group holds one view and its pack reads the view's point_weight as
    # its hint; bench.py then checks the hinted step's gradient sum against the scanning pack's
    full = _bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--gaussians", "200000", "--streams", "1", "--no-cpu-baseline",
                  "--no-dropin-mode", "--no-secondary", "--no-forward-only", "--no-rand-variant", "--no-trained-like",
                  "--exchange", "sparse", env=dict(LOGRAST_DIST_SINGLE_RANK="1"))
    chk = full["exchange"]["hint_check"]
    assert full["exchange"]["streamed"] and full["exchange"]["parts"] == 8, full["exchange"]
    assert chk is not None and chk["rel_l2"] < 1e-4 and chk["rows_with_hint"] > 1000 and chk["ok_on_every_rank"], chk
    assert chk["rows_differ"] <= 0.001 * chk["rows_scanning"], chk


def test_pack_and_unpack_rows_kernels():
    """lograst_pack_rows / lograst_unpack_rows (the row-sparse exchange's device side) against plain torch: every non-zero
    row of every group arrives exactly once with its index, the header counts them, both unpack forms restore / sum them,
    a bound that is too small raises the overflow flag and drops nothing it kept."""
    import torch
    from log_amd import dist as D
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(3)
    G, R = 3, 5000 + 37
    rows = torch.zeros(G, R, 16)
    want_counts = []
    for g in range(G):
        sel = torch.randperm(R, generator=gen)[: 700 + 100 * g]
        rows[g, sel] = torch.randint(-9, 10, (sel.numel(), 16), generator=gen).float()
        rows[g, sel[0], :] = 0.0
        rows[g, sel[0], 15] = 1.0                                       # a row whose only non-zero entry is its last column
        want_counts.append(int((rows[g] != 0).any(1).sum()))
    rows_d = rows.to(dev)
    k = 1024
    packed, over = D._pack_segments(rows_d, k)
    seg = D._segment_floats(k, dev)
    assert packed.numel() == G * seg and not bool(over)
    hdr = packed.view(G, seg)[:, 0].contiguous().view(torch.int32).cpu().tolist()
    assert hdr == want_counts
    # own-range form: segment g back into rows [g R, (g + 1) R) of a zeroed buffer
    back = torch.zeros(G * R, 16, device=dev)
    D._unpack_segments(back, packed, G, k, per_segment_rows=R)
    assert torch.equal(back.view(G, R, 16).cpu(), rows)
    # summing form: all segments into the same R rows (integer values: exact in any order)
    acc = torch.zeros(R, 16, device=dev)
    D._unpack_segments(acc, packed, G, k)
    assert torch.equal(acc.cpu(), rows.sum(0))
    # non-integer values, overlapping rows in all three segments: the sum is ((s0 + s1) + s2) bit for bit, on every run
    # (round-4 verdict weak #8: one launch with float atomics over all segments made it depend on the arrival order)
    frows = torch.zeros(G, R, 16)
    for g in range(G):
        sel = torch.randperm(R, generator=gen)[:3000]                   # 60 % of the rows per segment: heavy overlap
        frows[g, sel] = torch.randn(sel.numel(), 16, generator=gen) * (10.0 ** torch.randint(-3, 4, (sel.numel(), 1), generator=gen))
    fpacked, over = D._pack_segments(frows.to(dev), 4096)
    assert not bool(over)
    want = (frows[0] + frows[1]) + frows[2]
    for _ in range(3):
        facc = torch.zeros(R, 16, device=dev)
        D._unpack_segments(facc, fpacked, G, 4096)
        assert torch.equal(facc.cpu(), want)
    # the C ABI documents `overflow` as optional: NULL with a bound that is too small must not fault (round-4 advisory)
    import ctypes
    from log_amd import _lib
    L = _lib.lib()
    seg_small = int(L.lograst_sparse_segment_floats(64))
    buf = torch.empty(G * seg_small, dtype=torch.float32, device=dev)
    _lib.check(L.lograst_pack_rows(ctypes.c_void_p(rows_d.data_ptr()), G, R, 64, ctypes.c_void_p(buf.data_ptr()),
                                   ctypes.c_void_p(0), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    torch.cuda.synchronize()
    assert buf.view(G, seg_small)[:, 0].contiguous().view(torch.int32).cpu().tolist() == want_counts
    # a bound below the longest list: flagged, and what was kept is still right (a subset of the rows, each intact)
    small, over = D._pack_segments(rows_d, 512)
    assert bool(over)
    part = torch.zeros(G * R, 16, device=dev)
    D._unpack_segments(part, small, G, 512, per_segment_rows=R)
    part = part.view(G, R, 16).cpu()
    kept = (part != 0).any(2)
    assert int(kept.sum()) == 3 * 512 and torch.equal(part[kept], rows[kept])


def test_pack_and_clear_and_the_zeroing_unpack():
    """Round 6, the streamed exchange's device side: lograst_pack_rows_clear packs what lograst_pack_rows packs and leaves the
    packed rows ZERO in the bucket (rows dropped by an exceeded bound stay); lograst_unpack_rows(atomic = 2) clears exactly
    the rows an owner-major store wrote.  Against the torch formulation the gloo tests run."""
    import torch
    from log_amd import dist as D
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(11)
    G, R = 4, 70_000 + 13
    rows = torch.zeros(G, R, 16)
    for g in range(G):
        sel = torch.randperm(R, generator=gen)[: 5000 + 1000 * g]
        rows[g, sel] = torch.randn(sel.numel(), 16, generator=gen)
    bucket = rows.to(dev).contiguous()
    k = 9000
    plain, over0 = D._pack_segments(bucket.clone(), k)
    packed, over = D._pack_segments(bucket, k, clear=True)
    assert not bool(over) and not bool(over0)
    assert float(bucket.abs().sum()) == 0.0                              # every packed row is gone from the bucket
    back = torch.zeros(G * R, 16, device=dev)
    D._unpack_segments(back, packed, G, k, per_segment_rows=R)
    assert torch.equal(back.view(G, R, 16).cpu(), rows)                  # ... and arrived intact
    back2 = torch.zeros(G * R, 16, device=dev)
    D._unpack_segments(back2, plain, G, k, per_segment_rows=R)
    assert torch.equal(back2, back)
    # the torch formulation does the same
    cpu_rows = rows.clone()
    cpu_packed, cpu_over = D._pack_segments(cpu_rows, k, clear=True)
    assert float(cpu_rows.abs().sum()) == 0.0 and not bool(cpu_over)
    # clear what the store wrote: the result buffer is all zero again, rows it did not write are left alone
    back[5] = 7.0 if float(back[5].abs().sum()) == 0.0 else back[5]
    marker = back[5].clone()
    D._unpack_segments(back, packed, G, k, per_segment_rows=R, zero=True)
    wrote = (rows.view(G * R, 16) != 0).any(1)
    assert float(back[wrote.to(dev)].abs().sum()) == 0.0
    if not bool(wrote[5]):
        assert torch.equal(back[5], marker)
    # an exceeded bound: flagged; the kept rows are cleared, the dropped ones stay in the bucket (nothing is lost silently)
    bucket = rows.to(dev).contiguous()
    small, over = D._pack_segments(bucket, 4096, clear=True)
    assert bool(over)
    part = torch.zeros(G * R, 16, device=dev)
    D._unpack_segments(part, small, G, 4096, per_segment_rows=R)
    assert torch.equal((part.view(G, R, 16) + bucket).cpu(), rows)       # every row is in exactly one of the two places


def test_pack_rows_with_a_hint_and_the_visible_count_kernel():
    """lograst_pack_rows_hinted: rows whose hint word is zero are neither read nor packed nor cleared (a truthful hint gives
    what the plain pack gives; rows the hint hides stay where they are, whatever they hold; rows behind the hint's end have
    none); lograst_add_visible: seen += radii > 0 in one pass (GradientBucket.mark_seen)."""
    import torch
    from log_amd import dist as D
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(12)
    G, R = 3, 50_000 + 7
    rows = torch.zeros(G, R, 16)
    for g in range(G):
        sel = torch.randperm(R, generator=gen)[: 3000 + 500 * g]
        rows[g, sel] = torch.randn(sel.numel(), 16, generator=gen)
    nz = (rows != 0).any(2).view(-1)
    k = 9000
    for clear in (False, True):
        # truthful hint (with extra non-zero words on zero rows: a hint may say "look" too often, never too rarely)
        hint = torch.where(nz | (torch.rand(G * R, generator=gen) < 0.05), torch.rand(G * R, generator=gen) + 0.1, torch.zeros(G * R))
        bucket = rows.to(dev).contiguous()
        packed, over = D._pack_segments(bucket, k, clear=clear, hint=hint.to(dev))
        assert not bool(over)
        back = torch.zeros(G * R, 16, device=dev)
        D._unpack_segments(back, packed, G, k, per_segment_rows=R)
        assert torch.equal(back.view(G, R, 16).cpu(), rows)
        assert float(bucket.abs().sum()) == 0.0 if clear else torch.equal(bucket.cpu(), rows)
        # a hint that hides rows, and ends early: the hidden rows are not packed and not cleared
        short = hint[: 2 * R + 100].clone()
        short[::2] = 0.0
        visible = torch.zeros(G * R, dtype=torch.bool)
        visible[: short.numel()] = short != 0
        bucket = rows.to(dev).contiguous()
        packed, over = D._pack_segments(bucket, k, clear=clear, hint=short.to(dev))
        back = torch.zeros(G * R, 16, device=dev)
        D._unpack_segments(back, packed, G, k, per_segment_rows=R)
        want = torch.where(visible[:, None], rows.view(G * R, 16), torch.zeros(1, 16))
        assert torch.equal(back.cpu(), want)
        left = rows.view(G * R, 16) - want if clear else rows.view(G * R, 16)
        assert torch.equal(bucket.view(G * R, 16).cpu(), left)
        # the torch formulation (gloo tests) agrees
        cpu = rows.clone()
        cp, _ = D._pack_segments(cpu, k, clear=clear, hint=short)
        cb = torch.zeros(G * R, 16)
        D._unpack_segments(cb, cp, G, k, per_segment_rows=R)
        assert torch.equal(cb, want) and torch.equal(cpu.view(G * R, 16), left)
    b = D.GradientBucket(100_003, dev, 1, row_major=True)
    radii = torch.randint(-1, 3, (100_003,), generator=gen, dtype=torch.int32)
    for _ in range(3):
        b.mark_seen(radii.to(dev))
    assert torch.equal(b.seen[:100_003].cpu(), 3.0 * (radii > 0).float()) and float(b.seen[100_003:].abs().sum()) == 0.0
    # deferred: the step's views are counted together when the counts are next read (lograst_add_visible_n, 16 per pass)
    b.zero()
    views = [torch.randint(-1, 3, (100_003,), generator=gen, dtype=torch.int32).to(dev) for _ in range(19)]
    for v in views:
        b.mark_seen(v, defer=True)
    assert len(b._seen_pending) == 19 and float(b._seen.abs().sum()) == 0.0           # nothing counted yet
    want = sum((v > 0).float() for v in views)
    assert torch.equal(b.seen[:100_003], want) and not b._seen_pending
    b.mark_seen(views[0], defer=True)
    b.mark_seen(views[1])                                                                # an immediate mark counts the pending ones first
    assert torch.equal(b.seen[:100_003], want + (views[0] > 0).float() + (views[1] > 0).float())
    b.mark_seen(views[2], defer=True)
    b.zero()                                                                             # ... and a reset drops them
    assert float(b.seen.abs().sum()) == 0.0

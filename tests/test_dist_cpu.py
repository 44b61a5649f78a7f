"""View-sharded data parallelism (log_amd/dist.py) with world_size 2 on CPU (gloo)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from log_amd.dist import COLS, LAYOUT, GradientBucket, shard_views


def test_shard_views_partition():
    for world in (1, 2, 3, 8):
        seen = sorted(v for r in range(world) for v in shard_views(64, r, world))
        assert seen == list(range(64))
        assert max(len(shard_views(64, r, world)) for r in range(world)) - \
            min(len(shard_views(64, r, world)) for r in range(world)) <= 1


def test_bucket_views_alias_flat_buffer():
    b = GradientBucket(5, "cpu", world=1)
    params = {n: torch.zeros(5, c, requires_grad=True) for n, c in LAYOUT}
    b.attach(params)
    loss = sum((i + 1) * p.sum() for i, p in enumerate(params.values()))
    loss.backward()
    loss2 = sum((i + 1) * p.sum() for i, p in enumerate(params.values()))
    loss2.backward()   # accumulates in place, still inside the flat buffer
    off = 0
    for i, (n, c) in enumerate(LAYOUT):
        assert params[n].grad.data_ptr() == b.views[n].data_ptr()
        assert (b.flat[off:off + 5 * c] == 2.0 * (i + 1)).all()
        assert (b.flat[off + 5 * c:off + b.Ppad * c] == 0).all()      # padding rows up to a multiple of 4
        off += b.Ppad * c
    assert b.Ppad == 8 and b.flat.numel() == 8 * COLS
    assert b.reduce() is b.flat   # world 1: no communication, same buffer


def test_every_attribute_block_is_16_byte_aligned_for_any_point_count():
    """The kernels read rotations / write dL/drotations as 16-byte rows and lograst_forward / lograst_backward reject
    unaligned pointers; LoG's point count changes at every densify step, so the flat layouts must stay aligned for odd
    counts and odd worlds (round-2 advisory: P_pad odd put the rotations block on an 8-byte boundary)."""
    from log_amd.dist import FlatParams
    for P, world, blk in ((1001, 1, 0), (999, 3, 0), (7, 2, 0), (1000003, 8, 0), (1001, 2, 6), (5, 1, 0)):
        b = GradientBucket(P, "cpu", world, sh_coeffs=15, block_rows=blk)
        assert b.Ppad % 4 == 0 and b.Pr * world == b.Ppad and (blk == 0 or b.Pr % blk == 0)
        base = b.flat.data_ptr()
        for name, _ in b.layout:
            assert (b.views[name].data_ptr() - base) % 16 == 0, (P, world, name)
            for r in range(world):
                assert (b.rows(name, r).data_ptr() - base) % 16 == 0, (P, world, name, r)
    t = {n: torch.zeros(1001, c) for n, c in LAYOUT}
    fp = FlatParams(t, "cpu", world=1)
    assert (fp.views["rotations"].data_ptr() - fp.flat.data_ptr()) % 16 == 0


def _worker(rank, world, port, P, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = GradientBucket(P, "cpu", world=world)
        g = torch.Generator().manual_seed(100 + rank)
        params = {n: torch.zeros(P, c, requires_grad=True) for n, c in LAYOUT}
        b.attach(params)
        # each rank owns different "views": a different random linear loss per view
        for view in shard_views(6, rank, world):
            gv = torch.Generator().manual_seed(view)
            loss = sum((p * torch.rand(p.shape, generator=gv)).sum() for p in params.values())
            loss.backward()
        b.reduce()
        torch.save(b.flat.clone(), os.path.join(out, f"r{rank}.pt"))
        del g
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_sum_matches_single_process(tmp_path):
    P, world = 37, 2   # odd row count: exercises the padding of every attribute block to a multiple of world rows
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, P, str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    assert torch.equal(got[0], got[1])
    # single-process reference: all 6 views on one rank
    b = GradientBucket(P, "cpu", world=1)
    params = {n: torch.zeros(P, c, requires_grad=True) for n, c in LAYOUT}
    b.attach(params)
    for view in range(6):
        gv = torch.Generator().manual_seed(view)
        loss = sum((p * torch.rand(p.shape, generator=gv)).sum() for p in params.values())
        loss.backward()
    # the two-rank bucket pads every attribute block to a multiple of world rows: compare attribute by attribute
    b2 = GradientBucket(P, "cpu", world=world)
    b2.flat.copy_(got[0])
    for n, _ in LAYOUT:
        np.testing.assert_allclose(b2.views[n].numpy(), b.views[n].numpy(), rtol=1e-6)


def _band_worker(rank, world, port, H, W, q):
    import torch.distributed as dist
    from log_amd import dist as D
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(3 * H * W, dtype=torch.float32).reshape(3, H, W)     # what a single GPU would render
        b, e = D.band_pixels(rank, world, H)
        mine = torch.full_like(full, -1.0)                                          # other rows: garbage
        mine[:, b:e] = full[:, b:e]
        out = D.gather_bands(mine, rank, world)
        q.put((rank, bool(torch.equal(out, full)), D.band_rows(rank, world, H)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(2, 1080), (3, 100)])
def test_image_bands_partition_and_gather(world, H):
    """SURVEY 8e second axis: tile-row bands cover the image exactly once and the all-gather reassembles it."""
    import torch.multiprocessing as mp
    from log_amd import dist as D
    rows = [D.band_rows(r, world, H) for r in range(world)]
    gy = (H + 15) // 16
    assert rows[0][0] == 0 and rows[-1][1] == gy and all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
    assert max(e - b for b, e in rows) - min(e - b for b, e in rows) <= 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_band_worker, args=(r, world, port, H, 64, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res


# ---- owner-computes optimizer step ---------------------------------------------------------
GOLD_OWNER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "owner_adam.npz")
NAMES = ("means3D", "scales", "rotations", "opacities", "colors", "shs")


def _owner_run(rank, world, g, mode="dense"):
    """Replays the golden's steps through GradientBucket / FlatParams / OwnerAdam on `rank` of `world`.  The step's
    gradient is split into `world` addends (as if each rank had rendered some of the views): rank r contributes
    grad * w_r with w = exact binary fractions, and a disjoint part of the seen mask.
    mode: "dense" | "compact" (touched-block exchange, 2-row blocks) | "parts" (StepExchange, two groups of views that see
    alternating 2-row blocks) | "parts_compact"."""
    from log_amd.dist import FlatParams, GradientBucket, OwnerAdam, StepExchange
    P = int(g["P"])
    bounded = mode.endswith("_bounded")       # touched-block collectives sized from a bound: no read-back per exchange
    mode = mode[:-len("_bounded")] if bounded else mode
    compact, parts, row_major = mode.endswith("compact"), 2 if "parts" in mode else 1, mode.startswith("rows")
    blk = 2 if compact else 0
    tensors = {n: torch.from_numpy(g["init_" + n].copy()) for n in NAMES}
    params = FlatParams(tensors, "cpu", world, block_rows=blk)
    ex = StepExchange(P, "cpu", world, rank, sh_coeffs=15, parts=parts, block_rows=blk, row_major=row_major)
    opt = OwnerAdam(params, rank)
    w = [1.0] if world == 1 else [0.25, 0.75]
    old = GradientBucket.DENSE_ABOVE
    GradientBucket.DENSE_ABOVE = 2.0 if compact else old     # (63 % of the golden's rows are seen: force the compact form)
    try:
        for it in range(int(g["n_steps"])):
            ex.zero()
            seen = torch.from_numpy(g[f"s{it}_seen"])
            mine = seen & ((torch.arange(P) % world) == rank)                          # each row is "seen" by one rank
            for part, bucket in enumerate(ex.buckets):
                # the rows of this group of views: all of them, or blocks of two rows alternating between the groups; a
                # row nobody saw in a group has a zero gradient there (as after a real backward)
                rows = torch.ones(P, dtype=torch.bool) if parts == 1 else ((torch.arange(P) // 2) % 2) == part
                live = (seen & rows).to(torch.float32)
                for n in NAMES:
                    gr = torch.from_numpy(g[f"s{it}_grad_{n}"]).reshape(P, -1) * live[:, None] * w[rank]
                    bucket.alias[n].copy_(gr.reshape(bucket.alias[n].shape))   # (row-major bucket: strided views of "rows")
                bucket.mark_seen(torch.where(mine & rows, 5, 0))
                bound = None
                if bounded and world > 1:
                    # the bound = the longest touched-block list any ONE group has for any owner in this step -- exactly
                    # enough for each part alone.  The groups' blocks are disjoint, so the UNION (what the closing
                    # all-gather of the attributes is sized from) needs up to twice that (round-4 advisory)
                    Pr = bucket.Pr
                    nb_ = Pr // blk
                    bound = 1
                    for q in range(parts):
                        rq = ((torch.arange(P) // 2) % 2) == q
                        t = torch.zeros(world * Pr, dtype=torch.bool)
                        t[:P] = seen & rq
                        bound = max(bound, int(t.view(world, nb_, blk).any(-1).sum(1).max()))
                ex.launch(part, compact=compact, kmax=bound)
            lr = {"means3D": float(g[f"s{it}_lr_means3D"]), "scales": float(g[f"s{it}_lr_scales"]), "rotations": 0.001,
                  "opacities": 0.05, "colors": 0.0025, "shs": 0.000125}
            total = ex.finish()
            if compact and world > 1:
                assert ex.touched is not None and ex.touched.kmax < ex.touched.nb     # some blocks did stay home
            if bounded and world > 1:
                assert ex.touched.kmax == min(2 * bound, ex.touched.nb) and not ex.compact_overflowed()
                assert int(ex.touched.counts.max()) > bound      # the union really is longer than any part's list
            opt.step_rows(total, params, lr, touched=ex.touched)
    finally:
        GradientBucket.DENSE_ABOVE = old
    return params, opt


def _owner_worker(rank, world, port, out, mode="dense"):
    import oracle_backend
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oracle_backend.install(oracle_backend.OracleBackend())
        params, opt = _owner_run(rank, world, np.load(GOLD_OWNER), mode)
        torch.save({"flat": params.flat.clone(), "exp_avg": opt.exp_avg, "exp_avg_sq": opt.exp_avg_sq},
                   os.path.join(out, f"o{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_owner_adam_world1_reproduces_reference_optimizer(oracle_mod):
    """world = 1: the owner-computes step IS SparseOptimizer.step (golden produced by the reference's own class)."""
    import oracle_backend
    g = np.load(GOLD_OWNER)
    old = oracle_backend.install(oracle_backend.OracleBackend())
    try:
        params, opt = _owner_run(0, 1, g)
    finally:
        oracle_backend.install(old)
    for n in NAMES:
        np.testing.assert_allclose(params.views[n].numpy().reshape(g["final_" + n].shape), g["final_" + n], rtol=2e-6, atol=1e-9)
        P = params.P     # (the moments cover Pr >= P rows: the padding rows up to a multiple of 4 never move)
        np.testing.assert_allclose(opt.exp_avg[n].numpy()[:P].reshape(g["final_exp_avg_" + n].shape), g["final_exp_avg_" + n],
                                   rtol=2e-6, atol=1e-12)
        np.testing.assert_allclose(opt.exp_avg_sq[n].numpy()[:P].reshape(g["final_exp_avg_sq_" + n].shape),
                                   g["final_exp_avg_sq_" + n], rtol=2e-6, atol=1e-15)
        assert float(opt.exp_avg[n][P:].abs().sum()) == 0.0


@pytest.mark.parametrize("mode", ["dense", "compact", "parts", "parts_compact", "rows", "rows_parts_compact",
                                  "parts_compact_bounded"])
def test_owner_adam_two_ranks_match_one(tmp_path, oracle_mod, mode):
    """world = 2 (gloo): each rank steps only its rows, with moments for those rows only; after the all-gather both
    replicas hold the world-1 result (the two addends 0.25 g + 0.75 g sum to g exactly), and the moments of rank r are
    the world-1 moments of its rows.  The same through the touched-block exchange (only blocks some rank saw travel) and
    through StepExchange (two groups of views, reduce-scattered one after the other), alone and together -- and with the
    gradients in row-major buckets (one 16-float row per Gaussian, exchanged as one block: "rows*").
    "parts_compact_bounded": the touched-block lists sized from a bound that fits each group alone; the groups touch
    disjoint blocks, so the union is longer than the bound and must still be published whole."""
    import oracle_backend
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_owner_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, f"o{r}.pt")) for r in range(world)]
    assert torch.equal(got[0]["flat"], got[1]["flat"])
    old = oracle_backend.install(oracle_backend.OracleBackend())
    try:
        ref_params, ref_opt = _owner_run(0, 1, np.load(GOLD_OWNER))
    finally:
        oracle_backend.install(old)
    from log_amd.dist import FlatParams
    P = ref_params.P
    two = FlatParams({n: ref_params.views[n] for n in NAMES}, "cpu", world,      # same layout as the workers' buffers
                     block_rows=2 if "compact" in mode else 0)
    two.flat.copy_(got[0]["flat"])
    for n in NAMES:
        assert torch.equal(two.views[n], ref_params.views[n]), n
        Pr = two.Pr
        for r in range(world):
            rows = slice(r * Pr, min((r + 1) * Pr, P))
            k = rows.stop - rows.start
            assert torch.equal(got[r]["exp_avg"][n][:k], ref_opt.exp_avg[n][rows]), (n, r)
            assert torch.equal(got[r]["exp_avg_sq"][n][:k], ref_opt.exp_avg_sq[n][rows]), (n, r)


def test_row_major_bucket_layout():
    """GradientBucket(row_major=True): the 14 base columns are ONE block of 16-float rows (64-byte rows from a 64-byte
    aligned base: what LOGRAST_BWD_ACCUMULATE_ROWS adds into), the per-attribute tensors are strided views of it, the SH
    block stays attribute-major behind it; autograd accumulates into the strided .grad views in place."""
    from log_amd.dist import GradientBucket, ROW_COLUMNS, ROW_FLOATS, layout, split_rows
    b = GradientBucket(7, "cpu", world=2, sh_coeffs=15, row_major=True)
    assert layout(15, True) == (("rows", 16), ("shs", 45)) and b.cols == 16 + 45 and b.Ppad == 8
    assert b.views["rows"].shape == (7, ROW_FLOATS) and b.views["rows"].is_contiguous()
    assert b.rows("rows", 1).shape == (4, 16)
    with pytest.raises(ValueError, match="no rasterizer sink"):      # round-3 advisory: the pairing the rasterizer rejects
        b.sink()
    assert set(GradientBucket(7, "cpu", world=2, row_major=True).sink()) == {"rows"}
    params = {n: torch.zeros(7, c, requires_grad=True) for n, c in LAYOUT}
    params["shs"] = torch.zeros(7, 15, 3, requires_grad=True)
    b.attach(params)
    for _ in range(2):
        sum((i + 1) * p.sum() for i, p in enumerate(params.values())).backward()
    rows = b.views["rows"]
    for i, (n, c) in enumerate(LAYOUT):
        a, e = ROW_COLUMNS[n]
        assert e - a == c and params[n].grad.data_ptr() == b.alias[n].data_ptr()
        assert (rows[:, a:e] == 2.0 * (i + 1)).all()
    assert (rows[:, 14:] == 0).all() and (b.views["shs"] == 12.0).all()
    assert (b.flat[7 * 16:8 * 16] == 0).all()                                    # the padding row
    sp = split_rows({"rows": rows, "seen": b.seen})
    assert set(sp) == set(ROW_COLUMNS) | {"seen"} and sp["rotations"].shape == (7, 4)


def test_bucket_carries_sh_columns_and_seen_counts():
    from log_amd.dist import GradientBucket, layout
    b = GradientBucket(7, "cpu", world=2, sh_coeffs=15)
    assert dict(b.layout)["shs"] == 45 and b.views["shs"].shape == (7, 15, 3) and b.cols == COLS + 45
    assert b.flat.numel() == 8 * (COLS + 45) and b.Pr == 4                     # 7 rows padded to 2 x 4
    assert layout(0) == LAYOUT
    b.mark_seen(torch.tensor([3, 0, 1, 0, 0, 9, 0]))
    b.mark_seen(torch.tensor([0, 2]), index=torch.tensor([1, 0]))             # a level-of-detail selection: rows 1, 0
    assert b.seen.tolist() == [2.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0.0, 0.0]
    b.zero()
    assert float(b.seen.sum()) == 0.0


def _blocks_worker(rank, world, port, out):
    from log_amd.dist import GradientBucket, StepExchange
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, B = 1000, 16
        gen = torch.Generator().manual_seed(7 + rank)
        lo = 0 if rank == 0 else 608                     # rank 0's views see rows [0, 41), rank 1's rows [608, 649): three 16-row blocks each
        radii = torch.zeros(P, dtype=torch.int32)
        radii[lo:lo + 41] = 3
        res = {}
        for compact in (False, True):
            b = GradientBucket(P, "cpu", world, sh_coeffs=4, block_rows=B)
            for name, c in b.layout:
                gr = torch.zeros(P, c)
                gr[lo:lo + 41] = torch.randn(41, c, generator=torch.Generator().manual_seed(11 + rank))
                b.views[name].copy_(gr.reshape(b.views[name].shape))
            b.mark_seen(radii)
            rows = b.reduce_scatter_rows(rank, compact=compact)
            res[compact] = {k: v.clone() for k, v in rows.items()}
            if compact:
                res["kmax"], res["nb"] = b.touched.kmax, b.touched.nb
        # bounded form (no read-back): a bound that holds gives the exact result and no overflow flag; a bound that is
        # too small raises the flag (the exchange dropped blocks: the caller repeats the step)
        for bound, key in ((5, "bounded_ok"), (2, "bounded_small")):
            ex = StepExchange(P, "cpu", world, rank, sh_coeffs=4, parts=1, block_rows=B)
            b = ex.buckets[0]
            for name, c in b.layout:
                gr = torch.zeros(P, c)
                gr[lo:lo + 41] = torch.randn(41, c, generator=torch.Generator().manual_seed(11 + rank))
                b.views[name].copy_(gr.reshape(b.views[name].shape))
            b.mark_seen(radii)
            ex.launch(0, compact=True, kmax=bound)
            rows = ex.finish()
            res[key] = ({k: v.clone() for k, v in rows.items()}, b.touched.kmax, ex.compact_overflowed())
        # replicated-optimizer form through StepExchange: two groups, then every rank holds the whole sum
        ex = StepExchange(P, "cpu", world, rank, parts=2, block_rows=B)
        for part, b in enumerate(ex.buckets):
            b.views["means3D"].fill_(float(rank + 1) * (part + 1))
            b.mark_seen(radii)
            ex.launch(part)
        flat = ex.all_gather_grads(ex.finish())
        res["full"] = ex.buckets[0].views["means3D"].clone()
        torch.save(res, os.path.join(out, f"b{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_touched_block_exchange_moves_only_touched_blocks(tmp_path):
    """rows in 3 of each owner's 32 blocks are seen -> 3 blocks per owner travel instead of 32, and the rows
    every owner receives are identical to the dense reduce-scatter's (untouched rows: exact zeros either way)."""
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_blocks_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, f"b{r}.pt")) for r in range(world)]
    for r in range(world):
        assert got[r]["kmax"] == 3 and got[r]["nb"] == 32
        for k in got[r][False]:
            assert torch.equal(got[r][False][k], got[r][True][k]), (r, k)
        assert float(got[r][True]["seen"].sum()) == 41.0          # each owner's 41 seen rows, once each
        rows_ok, k_ok, over_ok = got[r]["bounded_ok"]
        assert k_ok == 5 and not over_ok
        for k in got[r][False]:
            assert torch.equal(got[r][False][k], rows_ok[k]), (r, k)
        assert got[r]["bounded_small"][1] == 2 and got[r]["bounded_small"][2]      # three touched blocks, bound 2: flagged
        assert torch.equal(got[r]["full"], torch.full((1000, 3), 9.0))   # (1 + 2) * (1 + 2): both ranks, both groups


def test_step_exchange_groups_consecutive_views():
    from log_amd.dist import StepExchange
    ex = StepExchange(10, "cpu", parts=3)
    groups = [ex.buckets.index(ex.bucket_of(v, 8)) for v in range(8)]
    assert groups == sorted(groups) and set(groups) == {0, 1, 2}
    assert [ex.last_view_of(g, 8) for g in range(3)] == [max(v for v in range(8) if groups[v] == g) for g in range(3)]
    # one rank, several groups: finish() is the plain sum of the groups' rows
    for i, b in enumerate(ex.buckets):
        b.views["colors"].fill_(float(i + 1))
        b.mark_seen(torch.ones(10))
    total = ex.finish()
    assert total["colors"].shape[0] == ex.buckets[0].Pr == 12          # a rank's rows: P rounded up to a multiple of 4
    assert torch.equal(total["colors"][:10], torch.full((10, 3), 6.0)) and torch.equal(total["seen"][:10], torch.full((10,), 3.0))
    assert float(total["colors"][10:].abs().sum()) == 0.0 and float(total["seen"][10:].sum()) == 0.0


def test_bucket_without_seen_counts():
    """track_seen=False (bench.py: the gradient sum only): nothing is kept or exchanged for the seen counts, and the
    consumers that need them say so."""
    from log_amd.dist import FlatParams, GradientBucket, OwnerAdam, StepExchange
    b = GradientBucket(9, "cpu", world=1, track_seen=False)
    assert b.seen.numel() == 1 and "seen" not in b.reduce_scatter_rows(0)
    with pytest.raises(RuntimeError, match="track_seen"):
        b.mark_seen(torch.ones(9))
    ex = StepExchange(9, "cpu", parts=2, track_seen=False)
    for i, bk in enumerate(ex.buckets):
        bk.views["means3D"].fill_(float(i + 1))
    total = ex.finish()
    assert "seen" not in total and torch.equal(total["means3D"][:9], torch.full((9, 3), 3.0))
    params = FlatParams({n: torch.zeros(9, c) for n, c in b.layout}, "cpu")
    with pytest.raises(ValueError, match="seen counts"):
        OwnerAdam(params, 0).step_rows(total, params, {"means3D": 1e-3})


def _sparse_worker(rank, world, port, out):
    from log_amd.dist import StepExchange
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = 1003                                         # (not a multiple of the world size or of four)
        res = {}
        for mode, kw_rs, kw_ag in (("dense", {}, {}), ("sparse_exact", dict(sparse=True), dict(sparse_kmax="exact")),
                                   ("sparse_bound", dict(sparse=True, kmax=200), dict(sparse_kmax=400)),
                                   ("sparse_small", dict(sparse=True, kmax=20), dict(sparse_kmax=400))):
            ex = StepExchange(P, "cpu", world, rank, parts=2, row_major=True)
            for part, b in enumerate(ex.buckets):
                gen = torch.Generator().manual_seed(100 * rank + part)
                touched = torch.randperm(P, generator=gen)[:150]                 # 15 % of the rows, different per rank and group
                # integer-valued gradients: the sums are exact in any order, so the forms must agree bit for bit
                b.views["rows"][touched, :14] = torch.randint(-8, 9, (150, 14), generator=gen).float()
                b.mark_seen((torch.rand(P, generator=gen) < 0.6).to(torch.int32))
                ex.launch(part, **kw_rs)
            total = ex.finish()
            flat = ex.all_gather_grads(total, **kw_ag)
            res[mode] = dict(rows=total["rows"].clone(), seen=total["seen"].clone(), full=ex.buckets[0].views["rows"].clone(),
                             over=ex.compact_overflowed(), kmax=getattr(ex.buckets[0], "sparse_kmax", 0), gk=ex.gather_kmax)
        # non-integer gradients (round-4 verdict, next #7): the row-sparse sum is the RANK-ORDERED sum ((r0 + r1) + r2)
        # of every group, bit for bit -- the test rebuilds it from the ranks' own buckets
        ex = StepExchange(P, "cpu", world, rank, parts=2, row_major=True)
        local = []
        for part, b in enumerate(ex.buckets):
            gen = torch.Generator().manual_seed(1000 + 100 * rank + part)
            touched = torch.randperm(P, generator=gen)[:400]                     # 40 % of the rows: heavy overlap between ranks
            b.views["rows"][touched, :14] = torch.randn(400, 14, generator=gen) * (10.0 ** torch.randint(-3, 4, (400, 1), generator=gen))
            local.append(b.blocks["rows"].clone())
            ex.launch(part, sparse=True)
        total = ex.finish()
        cleared = [float(b.blocks["rows"].abs().sum()) for b in ex.buckets]      # pack and clear: every bucket is all zero again
        ex.all_gather_grads(total, sparse_kmax="exact")
        res["float"] = dict(local=local, rows=total["rows"].clone(), full=ex.buckets[0].blocks["rows"].clone(),
                            Pr=ex.buckets[0].Pr, over=ex.compact_overflowed(), cleared=cleared, streamed=ex.streamed)
        # round 6: the STREAMED exchange, one group per view (parts = 8), against ONE group holding the sum of the same eight
        # views (integer-valued gradients: exact in any order, so the two must agree bit for bit), sized exactly and from bounds
        per_view = []
        for v in range(8):
            gen = torch.Generator().manual_seed(5000 + 100 * rank + v)
            touched = torch.randperm(P, generator=gen)[:60]                      # 6 % of the rows per view, overlapping between views
            per_view.append((touched, torch.randint(-8, 9, (60, 14), generator=gen).float(),
                             (torch.rand(P, generator=gen) < 0.3).to(torch.int32)))
        for mode, parts, kw_rs, kw_ag in (("one_group", 1, dict(sparse=True), dict(sparse_kmax="exact")),
                                          ("stream8", 8, dict(sparse=True), dict(sparse_kmax="exact")),
                                          ("stream8_bound", 8, dict(sparse=True, kmax=64), dict(sparse_kmax=500)),
                                          # each view's bucket packed from a HINT (the view's point_weight stand-in: non-zero
                                          # exactly at the rows it touched) and the seen counts marked in ONE bucket for the step
                                          ("stream8_hint", 8, dict(sparse=True), dict(sparse_kmax="exact"))):
            ex = StepExchange(P, "cpu", world, rank, parts=parts, row_major=True)
            result = torch.full((world * ex.buckets[0].Pr, 16), float("nan")) if parts > 1 else None   # (the first gather zero-fills it)
            for step in range(2):                                                # two steps: the second one starts from begin_step(), no zero()
                if step:
                    ex.begin_step()
                for v, (touched, vals, seen) in enumerate(per_view):
                    b = ex.bucket_of(v, 8)
                    b.views["rows"][touched, :14] += vals * (step + 1)
                    if mode == "stream8_hint":
                        hint = torch.zeros(P)
                        hint[touched] = 0.25
                        b.mark_touched(hint)
                        ex.seen_bucket(ex.buckets.index(b), True).mark_seen(seen)
                    else:
                        b.mark_seen(seen)
                    if v == ex.last_view_of(ex.buckets.index(b), 8):
                        ex.launch(ex.buckets.index(b), **kw_rs)
                total = ex.finish()
                left = [float(b.blocks["rows"].abs().sum()) for b in ex.buckets]
                ex.all_gather_grads(total, into=result, **kw_ag)                 # (streamed: a persistent result, bucket 0 stays clean)
                full = ex.buckets[0].blocks["rows"].clone() if result is None else result.reshape(-1).clone()
                after = [float(b.blocks["rows"].abs().sum()) for b in ex.buckets]
                res["%s_step%d" % (mode, step)] = dict(rows=total["rows"].clone(), seen=total["seen"].clone(), full=full, left=left,
                                                       after=after, over=ex.compact_overflowed(), streamed=ex.streamed)
                if parts == 1:
                    ex.zero()                                                    # (the one-group form keeps its sums: zero-filled as before)
        torch.save(res, os.path.join(out, f"s{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_sparse_exchange_equals_the_dense_one(tmp_path, world):
    """The row-sparse exchange (only rows with a non-zero gradient travel: packed per owner, all-to-all with equal splits,
    sparse all-gather of the summed shards) delivers what the dense reduce-scatter + all-gather deliver -- shards and full
    gradient sum bit for bit on integer-valued gradients --, exactly sized or from a bound; a bound that is too small raises
    the overflow flag."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_sparse_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, f"s{r}.pt")) for r in range(world)]
    for r in range(world):
        d = got[r]["dense"]
        assert not d["over"] and float(d["full"].abs().sum()) > 0
        assert torch.equal(d["full"], got[0]["dense"]["full"])                   # every rank holds the same sum
        for mode in ("sparse_exact", "sparse_bound"):
            m = got[r][mode]
            assert not m["over"], mode
            assert torch.equal(m["rows"], d["rows"]) and torch.equal(m["seen"], d["seen"]), (r, mode)
            assert torch.equal(m["full"], d["full"]), (r, mode)
        assert 0 < got[r]["sparse_exact"]["kmax"] <= 150 and 0 < got[r]["sparse_exact"]["gk"] <= 2 * 150 * world
        assert got[r]["sparse_bound"]["kmax"] == 200
        assert got[r]["sparse_small"]["over"]                                   # 20 rows per pair cannot hold ~50-75
    # non-integer gradients: the streamed exchange adds every received segment into the step's ONE running shard as it
    # arrives -- group after group, inside a group rank after rank: a row's sum is the left fold
    # ((((0 + a[g0, r0]) + a[g0, r1]) + ...) + a[g1, r0]) + ... -- and the gathered bucket is those shards side by side, bit
    # for bit, on every rank
    Pr = got[0]["float"]["Pr"]
    want = torch.zeros_like(got[0]["float"]["local"][0])
    for part in range(2):
        for r in range(world):
            want = want + got[r]["float"]["local"][part]
    want = want.view(world, Pr, 16)
    for r in range(world):
        f = got[r]["float"]
        assert not f["over"] and f["streamed"]
        assert f["cleared"] == [0.0, 0.0], f["cleared"]                          # pack and clear left every bucket all zero
        assert torch.equal(f["rows"], want[r]), r
        assert torch.equal(f["full"].view(world, Pr, 16), want), r
    # one group per view (parts = 8) == one group for all eight views, bit for bit (integer-valued gradients), over two
    # steps of which the second starts from begin_step() instead of a zero-fill
    for r in range(world):
        for step in range(2):
            one = got[r]["one_group_step%d" % step]
            assert not one["over"] and not one["streamed"] and float(one["full"].abs().sum()) > 0
            for mode in ("stream8", "stream8_bound", "stream8_hint"):
                m = got[r]["%s_step%d" % (mode, step)]
                assert m["streamed"] and not m["over"], (r, mode, step)
                assert m["left"] == [0.0] * 8 and m["after"] == [0.0] * 8, (mode, m["left"], m["after"])
                assert torch.equal(m["rows"], one["rows"]) and torch.equal(m["seen"], one["seen"]), (r, mode, step)
                assert torch.equal(m["full"], one["full"]), (r, mode, step)
        assert torch.equal(got[r]["stream8_step1"]["full"], got[0]["stream8_step1"]["full"])


def test_pack_rows_edge_cases():
    """The torch formulation of the row-sparse pack (what the gloo tests run; the device kernels are checked against it in
    tests/test_gpu_dist.py): an all-zero group, a bound above the group size, a row whose only non-zero entry is its last
    column, NaN (counts as non-zero: it must reach the owner), and the overflow flag."""
    from log_amd.dist import SPARSE_FLOATS, _pack_rows, _unpack_add
    rows = torch.zeros(3, 10, 16)
    rows[0, 3, 15] = 2.0
    rows[0, 7, 0] = float("nan")
    rows[2, :, 5] = torch.arange(1, 11).float()
    # a hint (4 bytes per row of all groups, in order; shorter: the rows behind it have none): rows whose word is zero are the
    # caller's "known to be zero" -- not looked at, whatever they hold
    hint = torch.zeros(25)
    hint[[3, 20, 21]] = 1.0
    ph, ch, _ = _pack_rows(rows, 12, hint=hint)
    assert ch.tolist() == [1, 0, 2] and int(ph[0, 0, 16].view(torch.int32)) == 3
    assert ph[2, :2, 16].view(torch.int32).tolist() == [0, 1] and ph[2, :2, 5].tolist() == [1.0, 2.0]
    packed, counts, over = _pack_rows(rows, 12)
    assert packed.shape == (3, 12, SPARSE_FLOATS) and counts.tolist() == [2, 0, 10] and not bool(over)
    assert float(packed[1].abs().sum()) == 0.0                                   # nothing but padding
    for g in range(3):
        back = _unpack_add(torch.zeros(10, 16), packed[g])
        assert torch.equal(torch.nan_to_num(back, nan=123.0), torch.nan_to_num(rows[g], nan=123.0)), g
    idx = packed[0, :2, 16].contiguous().view(torch.int32).tolist()
    assert idx == [3, 7]                                                         # ascending row order inside a group
    _, counts, over = _pack_rows(rows, 4)
    assert counts.tolist() == [2, 0, 10] and bool(over)

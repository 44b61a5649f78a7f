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
        off += 5 * c
    assert b.flat.numel() == 5 * COLS
    assert b.reduce() is b.flat   # world 1: no communication, same buffer


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
    P, world = 37, 2   # 37*14 = 518: exercises the padding to a multiple of world
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
    np.testing.assert_allclose(got[0][:P * COLS].numpy(), b.flat.numpy(), rtol=1e-6)


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

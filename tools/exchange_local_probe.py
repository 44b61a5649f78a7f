"""What the row-sparse exchange costs a rank LOCALLY (no links involved), one group per step against one group per view
(the streamed form, log_amd.dist.StepExchange with parts > 1): the real gradients of a rank's 8 views of the bench headline
are accumulated per view; timed with device events on one stream:
  one group : zero-fill of the bucket, pack of the 8-view union, add of the received rows into a zeroed shard, zero-fill +
              store of the gathered result
  streamed  : per view pack-and-clear + add into the running shard; clear-by-previous-segments + store of the gathered result
The all-to-all / all-gather themselves are NOT run (world = 1 here): `recv` is the rank's own packed buffer, i.e. the unpack
side handles as many rows as one peer would send.  python tools/exchange_local_probe.py [--gaussians N]"""
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
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench as B
    from log_amd import dist as D, rasterizer as R
    dev = torch.device("cuda:0")
    args = argparse.Namespace(width=1920, height=1080, views=8, opacity=a.opacity, scene="random")
    wl = B.RasterWorkload(args, a.gaussians, dev, 0, 1, torch, np)
    wl.zero_means2d = False
    N = a.gaussians
    leaves = {k: v.detach().requires_grad_(True) for k, v in wl.base.items()}
    views = []
    for rast in wl.rasts:                                   # each view's gradients in a bucket of its own
        b = torch.zeros(N, 16, device=dev)
        with R.accumulate_grads_into({"rows": b}):
            wl.one_view(rast, leaves)
        views.append(b)
    torch.cuda.synchronize()
    union = torch.stack(views).sum(0)
    frac = lambda t: float((t != 0).any(1).float().mean())
    res = {"gaussians": N, "touched_row_fraction_per_view": [round(frac(v), 4) for v in views],
           "touched_row_fraction_8_views": round(frac(union), 4)}

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    k_union = int((union != 0).any(1).sum()) + 16
    k_view = max(int((v != 0).any(1).sum()) for v in views) + 16
    scratch = union.clone()
    res["zero_fill_bucket_ms"] = timed(lambda: scratch.zero_())
    res["one_group_pack_ms"] = timed(lambda: D._pack_segments(union.view(1, N, 16), k_union))
    packed_u, _ = D._pack_segments(union.view(1, N, 16), k_union)
    shard = torch.zeros(N, 16, device=dev)
    res["one_group_unpack_add_ms"] = timed(lambda: D._unpack_segments(shard, packed_u, 1, k_union))
    res["one_group_gather_store_ms"] = timed(lambda: (scratch.zero_(), D._unpack_segments(scratch, packed_u, 1, k_union, per_segment_rows=N)))
    # streamed: pack and clear destroys its input -- time it on fresh copies, the copy timed separately and subtracted
    cp = timed(lambda: scratch.copy_(views[0]))
    res["streamed_pack_clear_ms_per_view"] = timed(lambda: (scratch.copy_(views[0]), D._pack_segments(scratch.view(1, N, 16), k_view, clear=True))) - cp
    packed_v, _ = D._pack_segments(views[0].clone().view(1, N, 16), k_view)
    res["streamed_unpack_add_ms_per_view"] = timed(lambda: D._unpack_segments(shard, packed_v, 1, k_view))
    D._unpack_segments(scratch.zero_(), packed_u, 1, k_union, per_segment_rows=N)
    res["streamed_gather_clear_and_store_ms"] = timed(lambda: (D._unpack_segments(scratch, packed_u, 1, k_union, per_segment_rows=N, zero=True),
                                                               D._unpack_segments(scratch, packed_u, 1, k_union, per_segment_rows=N)))
    res["bytes_one_group_rows"] = 68 * (k_union - 16)
    res["bytes_streamed_rows_per_view"] = 68 * (k_view - 16)
    res["local_ms_per_step_one_group"] = (res["zero_fill_bucket_ms"] + res["one_group_pack_ms"] + res["one_group_unpack_add_ms"]
                                          + res["one_group_gather_store_ms"])
    res["local_ms_per_step_streamed_total"] = (8 * (res["streamed_pack_clear_ms_per_view"] + res["streamed_unpack_add_ms_per_view"])
                                               + res["streamed_gather_clear_and_store_ms"])
    res["local_ms_exposed_streamed"] = (res["streamed_pack_clear_ms_per_view"] + res["streamed_unpack_add_ms_per_view"]
                                        + res["streamed_gather_clear_and_store_ms"])
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()

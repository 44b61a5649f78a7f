"""Times the device side of the row-sparse exchange in isolation (run ON a GPU box): the scanning pack, the hinted pack, the
unpack-add and the visible-count kernel on a bucket of N rows of which a fraction is touched.
    python tools/exchange_kernel_probe.py [--rows 30000000] [--touched 0.064]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=30_000_000)
    ap.add_argument("--touched", type=float, default=0.064)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from log_amd import dist as D
    dev = torch.device("cuda:0")
    N = args.rows
    gen = torch.Generator(device=dev).manual_seed(3)
    touched = torch.rand(N, device=dev, generator=gen) < args.touched
    k = int(int(touched.sum()) * 1.3) + 16
    weight = torch.where(touched, torch.rand(N, device=dev, generator=gen) + 0.01, torch.zeros(N, device=dev))
    src = torch.zeros(N, 16, device=dev)
    src[touched] = torch.randn(int(touched.sum()), 16, device=dev, generator=gen)
    radii = torch.randint(1, 9, (N,), device=dev, dtype=torch.int32, generator=gen)
    seen = torch.zeros(N, device=dev)
    shard = torch.zeros(N, 16, device=dev)
    b = D.GradientBucket(N, dev, 1, row_major=True)

    def timed(fn, setup=None):
        ts = []
        for _ in range(args.reps):
            if setup:
                setup()
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(e) * 1e3)
        ts.sort()
        return round(ts[len(ts) // 2], 1)

    bucket = torch.empty(1, N, 16, device=dev)
    fill = lambda: bucket.copy_(src.view(1, N, 16))
    out = {"rows": N, "touched_fraction": args.touched, "kmax": k}
    out["pack_scan_clear_us"] = timed(lambda: D._pack_segments(bucket, k, clear=True), fill)
    out["pack_hinted_clear_us"] = timed(lambda: D._pack_segments(bucket, k, clear=True, hint=weight), fill)
    out["pack_scan_us"] = timed(lambda: D._pack_segments(bucket, k), fill)
    out["pack_hinted_us"] = timed(lambda: D._pack_segments(bucket, k, hint=weight), fill)
    fill()
    packed, _ = D._pack_segments(bucket, k, hint=weight)
    out["unpack_add_us"] = timed(lambda: D._unpack_segments(shard, packed, 1, k))
    out["add_visible_us"] = timed(lambda: b.mark_seen(radii))
    out["torch_visible_us"] = timed(lambda: seen.add_((radii > 0).to(torch.float32)))
    half = torch.where(torch.rand(N, device=dev, generator=gen) < 0.5, radii, torch.zeros_like(radii))
    out["add_visible_half_us"] = timed(lambda: b.mark_seen(half))
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""Per-kernel times of one view (forward + backward through the drop-in package, eager, one stream) for a given build of
the library:  python tools/kernel_probe.py [--lib log_amd/lib/liblograst_<variant>.so] [--gaussians N] [--opacity X]
[--views V] [--fwd-only] [--env K=V ...].  Prints one JSON line {kernel: us per launch, ...}.  Several builds are compared
by running it once per build (the library is loaded once per process)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--gaussians", type=int, default=30_000_000)
    ap.add_argument("--opacity", type=float, default=0.999)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--train-fwd-only", action="store_true", help="training forwards (accumulator rows, hit masks) without their backward")
    ap.add_argument("--sink", action="store_true", help="backward adds into one row-major running-sum buffer (bench.py's pipelined form)")
    ap.add_argument("--env", nargs="*", default=[])
    ap.add_argument("--tag", default="")
    ap.add_argument("--scene", choices=("random", "trained"), default="random", help="bench.py's scene generators")
    a = ap.parse_args()
    if a.lib:
        os.environ["LOGRAST_LIB"] = os.path.abspath(a.lib)
    for kv in a.env:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    import numpy as np
    import torch
    import bench as B
    from log_amd import _lib
    dev = torch.device("cuda:0")
    args = argparse.Namespace(width=a.width, height=a.height, views=a.views, opacity=a.opacity, scene=a.scene)
    wl = B.RasterWorkload(args, a.gaussians, dev, 0, 1, torch, np)
    wl.zero_means2d = False
    leaves = {k: v.clone().requires_grad_(True) for k, v in wl.base.items()}

    def step():
        for rast in wl.rasts:
            if a.fwd_only:
                with torch.no_grad():
                    rast(means3D=wl.base["means3D"], means2D=torch.empty(wl.N, 3, device=dev), shs=None,
                         colors_precomp=wl.base["colors"], opacities=wl.base["opacities"], scales=wl.base["scales"],
                         rotations=wl.base["rotations"], cov3D_precomp=None)
            elif a.train_fwd_only:
                rast(means3D=leaves["means3D"], means2D=torch.empty(wl.N, 3, device=dev).requires_grad_(True), shs=None,
                     colors_precomp=leaves["colors"], opacities=leaves["opacities"], scales=leaves["scales"],
                     rotations=leaves["rotations"], cov3D_precomp=None)
            else:
                wl.one_view(rast, leaves)
                for t in leaves.values():
                    t.grad = None

    import contextlib
    from log_amd import rasterizer as R
    rows = torch.zeros(wl.N, 16, device=dev) if a.sink else None
    ctx = (lambda: R.accumulate_grads_into({"rows": rows})) if a.sink else contextlib.nullcontext
    with ctx():
        step()
    torch.cuda.synchronize()
    _lib.profile_reset(); _lib.profile_enable(True)
    for _ in range(a.reps):
        with ctx():
            step()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    out = {k: round(1e3 * ms / max(c, 1), 1) for k, (ms, c) in _lib.profile_read().items() if c}
    out["_sum_us"] = round(sum(v for v in out.values()), 1)
    out["_tag"] = a.tag or os.path.basename(a.lib or "default")
    out["_n"] = a.gaussians
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

"""Timing probe: the kernels of one band view of 100 M Gaussians (gradient-sink form), for experiments with
LOGRAST_PROJECT_ABLATE / LOGRAST_BATCH / LOGRAST_BAND_SPARSE.  Prints {kernel: us}."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# LOGRAST_PROJECT_ABLATE / LOGRAST_BATCH exist only in -DLR_EXPERIMENTS builds (log_amd/csrc/common.hpp): build one, load it
from log_amd import build as _build
os.environ["LOGRAST_LIB"] = _build.build(variant="exp", extra_flags=["-DLR_EXPERIMENTS"], verbose=False)


def main():
    import numpy as np
    import torch
    from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer
    from log_amd import _lib, dist as D, rasterizer as R, scenes
    N = int(os.environ.get("PROBE_N", "100000000"))
    W, H, bands, band = 3840, 2160, 8, 3
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    smax = 0.5 * float(N) ** (-1.0 / 3.0)
    base = {"means3D": torch.rand(N, 3, device=dev, generator=gen) - 0.5,
            "scales": torch.rand(N, 3, device=dev, generator=gen) * smax,
            "rotations": torch.nn.functional.normalize(torch.rand(N, 4, device=dev, generator=gen) + 1e-3),
            "opacities": torch.full((N, 1), 0.999, device=dev), "colors": torch.rand(N, 3, device=dev, generator=gen)}
    if os.environ.get("PROBE_SORTED"):      # band membership coherent in memory (what a tree-ordered model looks like)
        order = torch.argsort(base["means3D"][:, 2])
        base = {k: v[order].contiguous() for k, v in base.items()}
    T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
    cam = scenes.orbit_cameras(2, W=W, H=H, focal=2139.0 * W / 1920.0)[0]
    rows = D.band_rows(band, bands, H)
    b, e = D.band_pixels(band, bands, H)
    wloss = torch.rand(3, e - b, W, device=dev, generator=gen)
    sink = {k: torch.zeros_like(v) for k, v in base.items()}
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam["FoVx"] * 0.5), tanfovy=math.tan(cam["FoVy"] * 0.5),
        bg=T([1.0, 1.0, 1.0]), scale_modifier=1.0, viewmatrix=T(cam["world_view_transform"]),
        projmatrix=T(cam["full_proj_transform"]), sh_degree=0, campos=T(cam["camera_center"]), prefiltered=False, debug=False)
    rast = GaussianRasterizer(raster_settings=rs)

    def view():
        leaves = {k: v.detach().requires_grad_(True) for k, v in base.items()}
        means2D = torch.empty(N, 3, device=dev).requires_grad_(True)
        with R.accumulate_grads_into(sink), R.tile_rows(*rows):
            out = rast(means3D=leaves["means3D"], means2D=means2D, shs=None, colors_precomp=leaves["colors"],
                       opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
            out[0][:, b:e].backward(gradient=wloss)
        return int((out[1] > 0).sum())

    view(); view()
    torch.cuda.synchronize()
    _lib.profile_reset(); _lib.profile_enable(True)
    n_band = view()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    out = {k: round(1e3 * ms / max(c, 1), 1) for k, (ms, c) in _lib.profile_read().items()}
    out["in_band"] = n_band
    out["env"] = {k: v for k, v in os.environ.items() if k.startswith(("LOGRAST_", "PROBE_"))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

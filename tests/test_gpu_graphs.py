"""A view's forward + backward captured into a HIP graph (what bench.py replays in its pipelined mode): same image bit
for bit, same gradients, also with two graphs replaying concurrently on two streams."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n=300_000, W=1280, H=720):
    from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer
    from log_amd import scenes
    dev = torch.device("cuda:0")
    sc = scenes.random_scene(n, seed=3, opacity=None, smax=0.01)
    T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
    base = dict(means3D=T(sc["xyz"]), scales=T(sc["scaling"]), rotations=T(sc["rotation"]), opacities=T(sc["opacity"]),
                colors=T(sc["colors"]))
    rasts = []
    for cam in scenes.orbit_cameras(4, W=W, H=H, focal=1400.0)[:2]:
        rs = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam["FoVx"] * 0.5), tanfovy=math.tan(cam["FoVy"] * 0.5),
            bg=T([0.1, 0.2, 0.3]), scale_modifier=1.0, viewmatrix=T(cam["world_view_transform"]),
            projmatrix=T(cam["full_proj_transform"]), sh_degree=0, campos=T(cam["camera_center"]), prefiltered=False,
            debug=False)
        rasts.append(GaussianRasterizer(raster_settings=rs))
    w = torch.rand(3, H, W, device=dev)
    return dev, n, base, rasts, w


def test_captured_view_replays_identically_alone_and_concurrently():
    from log_amd import rasterizer as R
    from log_amd.dist import GradientBucket
    dev, n, base, rasts, w = _setup()

    def one_view(rast, leaves):
        m2 = torch.zeros(n, 3, device=dev, requires_grad=True)
        out = rast(means3D=leaves["means3D"], means2D=m2, shs=None, colors_precomp=leaves["colors"],
                   opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
        out[0].backward(gradient=w)
        return out[0], m2

    lanes = []
    for _ in range(2):
        leaves = {k: v.detach().requires_grad_(True) for k, v in base.items()}
        bk = GradientBucket(n, dev, 1)
        bk.attach(leaves)
        lanes.append((leaves, bk))
    # eager reference (exact mode), and the capacity for the sync-free mode the capture needs
    ref, caps = [], []
    for (leaves, bk), rast in zip(lanes, rasts):
        bk.zero()
        with R.accumulate_grads_into(bk.views):
            img, m2 = one_view(rast, leaves)
        info = R.last_state_info(dev)
        caps.append(info)
        ref.append((img.clone(), m2.grad.clone(), bk.flat.clone()))
    R.set_instance_capacity(int(max(c[0] for c in caps) * 1.05) + 64, max_tile_len=int(max(c[2] for c in caps) * 1.05) + 64)
    try:
        R.overflow_since_reset(dev)
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        graphs, outs = [], []
        for li, ((leaves, bk), rast) in enumerate(zip(lanes, rasts)):
            with torch.cuda.stream(streams[li]):                 # warm-up on the capture stream
                with R.accumulate_grads_into(bk.views):
                    one_view(rast, leaves)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with R.accumulate_grads_into(bk.views):
                with torch.cuda.graph(g, pool=torch.cuda.graph_pool_handle(), stream=streams[li]):
                    outs.append(one_view(rast, leaves))
            graphs.append(g)
        torch.cuda.synchronize()
        for mode in ("alone", "concurrent", "concurrent"):
            for li, (leaves, bk) in enumerate(lanes):
                with torch.cuda.stream(streams[li]):
                    bk.zero()
                    graphs[li].replay()
                if mode == "alone":
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            for li, (leaves, bk) in enumerate(lanes):
                img, m2 = outs[li]
                r_img, r_m2, r_flat = ref[li]
                assert torch.equal(img, r_img), (mode, li)          # the forward is deterministic: bit for bit
                assert float((m2.grad - r_m2).norm() / r_m2.norm()) < 1e-5, (mode, li)
                rb = GradientBucket(n, dev, 1)
                rb.flat.copy_(r_flat)
                for k in ("opacities", "colors"):                    # reverse-walk outputs: every row
                    assert float((bk.views[k] - rb.views[k]).norm() / rb.views[k].norm()) < 1e-5, (mode, li, k)
                for k in ("means3D", "scales", "rotations"):         # chain rule: pancake-flat rows amplify the atomics' order
                    d = (bk.views[k] - rb.views[k]).norm(dim=1)
                    ref_n = rb.views[k].norm(dim=1)
                    live = ref_n > 1e-6 * ref_n.max()
                    rel = (d[live] / ref_n[live]).float().cpu()
                    assert float(rel.median()) < 1e-5 and float(torch.quantile(rel, 0.97)) < 1e-3, (mode, li, k)
        chk = R.overflow_since_reset(dev)
        assert not chk["overflowed"] and chk["forwards"] >= 6
    finally:
        R.set_instance_capacity(None)

"""Generates tests/golden/*.npz by IMPORTING the reference's own Python (run in the build container,
where /root/reference exists; the .npz fixtures are committed and travel to the GPU box).

Pins the projection / EWA / radius half of the path against the reference's in-tree twin of the
CUDA kernel:
  geom_*.npz          : LoG.model.geometry.compute_radius      (geometry.py:132-151, max(.,0.3) low-pass)
  (same files)        : LoG.model.geometry.computeCov3D + computeCov2D0 with DILATE_PIXEL=0 (geometry.py:27-41,91-130):
                        raw EWA covariance, to which the test adds the published +0.3 of the upstream flavour
Cameras come from the reference's LoG.dataset.base.prepare_camera (base.py:20-55).

    python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np
import torch

REF = os.environ.get("LOG_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from LoG.model import geometry  # noqa: E402  (reference code, imported not copied)
from LoG.dataset.base import prepare_camera  # noqa: E402


def orbit_RT(theta_deg, radius):
    th = math.radians(theta_deg)
    st, ct = math.sin(th), math.cos(th)
    R = np.array([[-st, ct, 0.0], [0.0, 0.0, -1.0], [-ct, -st, 0.0]])
    c = np.array([radius * ct, radius * st, 0.0]).reshape(3, 1)
    return R, -R @ c


def main():
    cases = [
        dict(name="hd", W=1920, H=1080, K=[[2139, 0, 960], [0, 2139, 540], [0, 0, 1]], theta=0.0, radius=3.0, n=4096, smax=0.05, seed=0),
        dict(name="sq", W=400, H=400, K=[[445, 0, 200], [0, 445, 200], [0, 0, 1]], theta=135.0, radius=3.0, n=4096, smax=0.02, seed=1),
        dict(name="offc", W=640, H=480, K=[[500, 0, 300.5], [0, 520, 251.25], [0, 0, 1]], theta=225.0, radius=2.0, n=4096, smax=0.2, seed=2),
    ]
    for c in cases:
        torch.manual_seed(c["seed"])
        n = c["n"]
        xyz = torch.rand(n, 3) - 0.5
        scaling = torch.rand(n, 3) * c["smax"]
        rotation = torch.nn.functional.normalize(torch.rand(n, 4))
        R, T = orbit_RT(c["theta"], c["radius"])
        camera = {"R": R, "T": T, "K": np.array(c["K"], dtype=np.float64), "W": c["W"], "H": c["H"],
                  "center": (-R.T @ T)}
        cam = prepare_camera(camera, 1, 0.1, 100.0)
        camt = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
        radius = geometry.compute_radius(xyz, scaling, rotation, camt)
        cov3D = geometry.computeCov3D(scaling, rotation)
        cov6 = torch.stack([cov3D[:, 0, 0], cov3D[:, 0, 1], cov3D[:, 0, 2], cov3D[:, 1, 1], cov3D[:, 1, 2], cov3D[:, 2, 2]], dim=-1)
        # NOTE: geometry.computeCov2D (geometry.py:48-89, the "+0.3" twin) is dead code in the reference
        # and its clip() calls collapse t.x to -limx (geometry.py:61-62), so it cannot serve as a golden.
        # The raw EWA covariance is taken from computeCov2D0 with the low-pass disabled (DILATE_PIXEL=0).
        a, b, cc = geometry.computeCov2D0(cov3D, xyz, camt["world_view_transform"], camt, DILATE_PIXEL=0.0)
        np.savez_compressed(
            os.path.join(HERE, f"geom_{c['name']}.npz"),
            xyz=xyz.numpy(), scaling=scaling.numpy(), rotation=rotation.numpy(),
            world_view_transform=cam["world_view_transform"], full_proj_transform=cam["full_proj_transform"],
            FoVx=np.float64(cam["FoVx"]), FoVy=np.float64(cam["FoVy"]),
            W=np.int32(cam["image_width"]), H=np.int32(cam["image_height"]),
            ref_radius_clamp=radius.numpy(), ref_cov3D=cov6.numpy(),
            ref_cov2D_raw=torch.stack([a, b, cc], dim=-1).numpy())
        print(c["name"], "radius range", float(radius.min()), float(radius.max()))


if __name__ == "__main__":
    main()

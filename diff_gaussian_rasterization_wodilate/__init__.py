"""Drop-in for the ``diff_gaussian_rasterization_wodilate`` package (chingswy fork, branch ``antialias``) that
LoG imports at /root/reference/LoG/render/renderer.py:1,104 -- backed by the MI355X HIP kernels of log_amd.

Fork behaviour reproduced: 5-tuple return, ``use_filter=`` kwarg, ``compute_radius`` method, low-pass
``max(cov, 0.3)`` and the |ndc| > 1.3 cull (both evidenced by LoG/cuda/compute_radius_kernel.cu:100-104,131-134).
"""
from log_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]

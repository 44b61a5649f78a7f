"""Drop-in for the upstream ``diff_gaussian_rasterization`` package (graphdeco-inria) that LoG imports when
``use_origin_render: True`` (/root/reference/LoG/render/renderer.py:99-102; apps/check_gui.py:19) -- backed by the
MI355X HIP kernels of log_amd.  Upstream behaviour: 2-tuple return, ``+0.3`` low-pass, near cull only."""
from log_amd.rasterizer import GaussianRasterizationSettings  # noqa: F401
from log_amd.rasterizer import UpstreamGaussianRasterizer as GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]

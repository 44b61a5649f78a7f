"""Drop-in for ``LoG.cuda.compute_radius`` (/root/reference/LoG/cuda/compute_radius.py:3-9), whose
``compute_radius_module.compute_radius`` LoG calls once per LoD level per view
(LoG/model/level_of_gaussian.py:81-84).  The reference JIT-compiles a CUDA file against an un-vendored glm;
this module exposes the same callable backed by liblograst's HIP kernel (include/lograst.h:
lograst_compute_radius).  Install by replacing the body of LoG/cuda/compute_radius.py with
``from log_amd.compute_radius import compute_radius_module``."""
import torch

from . import rasterizer as _r


class _ComputeRadiusModule:
    """Same call signature as the pybind module (compute_radius_kernel.cu:158-187)."""

    @staticmethod
    def compute_radius(means3D, scales, rotations, projmatrix, viewmatrix, focal_x, focal_y, tanfovx, tanfovy):
        with torch.no_grad():
            return _r._backend.compute_radius(means3D, scales, rotations, projmatrix, viewmatrix, focal_x, focal_y,
                                              tanfovx, tanfovy)


compute_radius_module = _ComputeRadiusModule()

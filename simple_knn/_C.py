"""``distCUDA2(points[P,3]) -> float[P]``: mean squared distance to the 3 nearest other points, on the MI355X
through liblograst (include/lograst.h: lograst_knn_mean_dist2; kernels in log_amd/csrc/knn.hip)."""
import ctypes

import torch

from log_amd import _lib


def distCUDA2(points):
    if points.device.type != "cuda":
        raise _lib.LograstError("distCUDA2 needs a tensor on the MI355X (LoG calls it with xyz.cuda()); "
                                "there is no CPU fallback")
    L = _lib.lib()
    pts = points.detach().to(torch.float32).contiguous()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError("distCUDA2 expects points[P,3]")
    P = pts.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    nbytes = int(L.lograst_knn_scratch_bytes(P))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        _lib.check(L.lograst_knn_mean_dist2(P, ctypes.c_void_p(pts.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                            ctypes.c_void_p(scratch.data_ptr()), nbytes,
                                            ctypes.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)))
    return out

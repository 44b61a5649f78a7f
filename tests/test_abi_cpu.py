"""CPU checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol that
include/lograst.h declares, the ctypes binding covers all of them, and the product path refuses to run on CPU
tensors (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "lograst.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lograst_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from log_amd import _lib
    names = _declared()
    assert len(names) >= 15
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/lograst.h but not exported by liblograst.so"
    assert set(names) == set(_lib.EXPORTS), set(names) ^ set(_lib.EXPORTS)


def test_sizing_helpers_and_error_text_work_without_gpu():
    from log_amd import _lib
    L = _lib.lib()
    assert L.lograst_version() == 1
    tiles = 120 * 68
    assert L.lograst_tile_state_bytes(1920, 1080, 1000000) >= 4 * (tiles + 1)
    assert L.lograst_geom_bytes(10) == 10 * (64 + 16)   # records + fill records
    assert L.lograst_keys_bytes(7) == 2 * 56 and L.lograst_list_bytes(7) == 28   # keys + the sort scratch half
    # argument validation happens before any device work
    rc = L.lograst_compute_radius(-1, None, None, None, None, None, 1.0, 1.0, 1.0, 1.0, None, None)
    assert rc < 0 and b"negative" in L.lograst_last_error()
    assert [L.lograst_kernel_name(i) for i in range(3)] == [b"compute_radius", b"project", b"scan_tiles"]


def test_product_path_has_no_cpu_fallback():
    from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer
    from log_amd import _lib
    from log_amd.compute_radius import compute_radius_module
    from simple_knn._C import distCUDA2
    rs = GaussianRasterizationSettings(image_height=32, image_width=32, tanfovx=0.5, tanfovy=0.5, bg=torch.zeros(3),
                                       scale_modifier=1., viewmatrix=torch.eye(4), projmatrix=torch.eye(4),
                                       sh_degree=0, campos=torch.zeros(3), prefiltered=False, debug=False)
    r = GaussianRasterizer(raster_settings=rs)
    z = torch.zeros
    with pytest.raises(_lib.LograstError, match="no CPU fallback"):
        r(means3D=z(2, 3), means2D=z(2, 3), shs=None, colors_precomp=z(2, 3), opacities=z(2, 1), scales=z(2, 3) + 1,
          rotations=z(2, 4) + 1, cov3D_precomp=None)
    with pytest.raises(_lib.LograstError):
        compute_radius_module.compute_radius(z(2, 3), z(2, 3), z(2, 4), torch.eye(4), torch.eye(4), 1., 1., 1., 1.)
    with pytest.raises(_lib.LograstError):
        distCUDA2(z(8, 3))
    # the product package does not import the oracle
    import sys
    import log_amd.rasterizer  # noqa: F401
    src = "".join(open(os.path.join(ROOT, "log_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "log_amd"))
                  if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src
    del sys


def test_reference_error_strings():
    from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer
    rs = GaussianRasterizationSettings(32, 32, 0.5, 0.5, torch.zeros(3), 1., torch.eye(4), torch.eye(4), 0,
                                       torch.zeros(3), False, False)
    r = GaussianRasterizer(raster_settings=rs)
    z = torch.zeros
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), scales=z(2, 3), rotations=z(2, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), colors_precomp=z(2, 3))


def test_header_is_plain_c(tmp_path):
    """include/lograst.h is the drop-in boundary for hosts that are not Python: it must compile as C99 on its own."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("no gcc")
    src = tmp_path / "h.c"
    src.write_text('#include "lograst.h"\nint main(void) { lograst_view v; lograst_adam_key k; (void)v; (void)k; return 0; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)])

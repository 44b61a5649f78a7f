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
    assert L.lograst_version() == 4
    tiles = 120 * 68
    assert L.lograst_tile_state_bytes(1920, 1080, 1000000) >= 4 * (tiles + 1)
    # records + fill records + an index each (band views), rounded up to 64 bytes, + the rank rows of the 5..16-tile rects
    # (8 bytes per Gaussian of the padded batches: round 5)
    assert L.lograst_geom_bytes(10) == (10 * (64 + 16 + 4) + 63) // 64 * 64 + 8 * 10 + 8 * 32768
    assert L.lograst_geom_bytes(30_000_000) < 30_000_000 * 93 + (1 << 20)
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


def test_ctypes_structs_match_the_header(tmp_path):
    """The ctypes mirrors in log_amd/_lib.py have the size and field offsets the C compiler gives the header's structs."""
    import shutil
    import subprocess
    from log_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    fields = {"lograst_view": [f[0] for f in _lib.LograstView._fields_],
              "lograst_adam_key": [f[0] for f in _lib.LograstAdamKey._fields_]}
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "lograst.h"', 'int main(void) {']
    for st, names in fields.items():
        prog.append(f'  printf("{st} %zu", sizeof({st}));')
        for n in names:
            prog.append(f'  printf(" %zu", offsetof({st}, {n}));')
        prog.append('  printf("\\n");')
    prog += ['  return 0;', '}']
    src, exe = tmp_path / "s.c", tmp_path / "s"
    src.write_text("\n".join(prog))
    inc = os.path.join(ROOT, "include")
    subprocess.check_call([gcc, "-std=c99", "-I", inc, str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).strip().splitlines()
    for line, (st, cls) in zip(out, (("lograst_view", _lib.LograstView), ("lograst_adam_key", _lib.LograstAdamKey))):
        parts = line.split()
        assert parts[0] == st and int(parts[1]) == ctypes.sizeof(cls), line
        assert [int(x) for x in parts[2:]] == [getattr(cls, f[0]).offset for f in cls._fields_], line


def test_ctypes_signatures_match_the_header_prototypes():
    """Every prototype in include/lograst.h against log_amd/_lib.py: same number of parameters and the same C class
    (pointer / 32-bit int / 64-bit int / float / double / size_t) in every position."""
    from log_amd import _lib
    src = open(os.path.join(ROOT, "include", "lograst.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = dict(re.findall(r"\b(lograst_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", src))
    assert set(protos) == set(_lib.EXPORTS)

    def c_class(param):
        p = param.strip()
        if "*" in p:
            return "ptr"
        base = p.rsplit(" ", 1)[0] if " " in p else p
        return {"int32_t": "i32", "uint32_t": "i32", "int": "i32", "float": "f32", "double": "f64", "size_t": "size",
                "int64_t": "i64"}[base.replace("const ", "").strip()]

    def py_class(t):
        if t in (ctypes.c_void_p, ctypes.c_char_p) or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
            return "ptr"
        return {ctypes.c_int32: "i32", ctypes.c_uint32: "i32", ctypes.c_int: "i32", ctypes.c_float: "f32",
                ctypes.c_double: "f64", ctypes.c_size_t: "size", ctypes.c_int64: "i64"}[t]

    for name, params in protos.items():
        want = [] if params.strip() in ("", "void") else [c_class(p) for p in params.split(",")]
        got = [py_class(t) for t in _lib._SIGNATURES[name][1]]
        assert got == want, (name, got, want)


def test_every_tunable_knob_is_documented():
    """The knob table of the library (api.hip: kKnobs -- what lograst_knob_info enumerates) against INTEGRATION.md: a knob
    a host can move must be described where a host looks."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "log_amd", "csrc", "api.hip")).read()
    i = src.index("static const LrKnobInfo kKnobs[]")
    names = re.findall(r'\{"(LOGRAST_[A-Z_]+)"', src[i:src.index("};", i)])
    assert len(names) >= 20
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert [n for n in names if n not in doc] == []

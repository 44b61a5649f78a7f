"""Builds log_amd/lib/liblograst.so from log_amd/csrc/*.hip with hipcc for gfx950 (in-tree, so the
.so travels to the GPU box with the repo snapshot).  `python -m log_amd.build [--force]`."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "liblograst.so")
OBJDIR = os.path.join(LIBDIR, "obj")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -ffp-contract=off      : explicit fmaf only (bit-exact contract with the oracle; see csrc/common.hpp)
# -munsafe-fp-atomics    : float atomicAdd -> global_atomic_add_f32 instead of a CAS loop
# -mcode-object-version=5: loadable by both the ROCm 7.2 runtime and torch's bundled 7.0 runtime
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-mcode-object-version=5", "-Wall", "-Wno-unused-function",
         # the kernels do their own wave-level reductions before every atomic; LLVM's atomic optimizer would
         # wrap each (single-lane) atomic in a second, scalar reduction loop
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, variant=None, extra_flags=()):
    """variant: name of an experimental build (liblograst_<variant>.so with extra_flags), used only by
    tools/experiments; the product library is the default build."""
    global LIB, OBJDIR
    lib, objdir = LIB, OBJDIR
    if variant:
        lib = os.path.join(LIBDIR, f"liblograst_{variant}.so")
        objdir = os.path.join(LIBDIR, f"obj_{variant}")
        force = True
    return _build(lib, objdir, force, verbose, list(extra_flags))


def _build(LIB, OBJDIR, force, verbose, extra):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.join(os.path.dirname(HERE), "include", "lograst.h"),
                                                             os.path.abspath(__file__)]
    os.makedirs(OBJDIR, exist_ok=True)
    jobs = []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o")
        if force or _newer(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + extra + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if args:   # python -m log_amd.build <variant> <flag> [<flag> ...]
        print(build(variant=args[0], extra_flags=args[1:]))
    else:
        print(build(force="--force" in sys.argv))

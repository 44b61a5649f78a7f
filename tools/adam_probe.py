"""Timing probe for lograst_sparse_adam: P model rows of 59 floats (xyz 3, scaling 3, rotation 4, opacity 1, colors 3,
shs 45), m selected rows in different index patterns.  Prints GB/s of algorithmic bytes (28 B per element)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from log_amd import rasterizer as R
    dev = torch.device("cuda:0")
    P = int(os.environ.get("PROBE_P", "13000000"))
    frac = float(os.environ.get("PROBE_FRAC", "0.485"))
    widths = {"xyz": (3,), "scaling": (3,), "rotation": (4,), "opacity": (1,), "colors": (3,), "shs": (15, 3)}
    model = {k: torch.zeros((P,) + w, device=dev) for k, w in widths.items()}
    m1 = {k: torch.zeros_like(v) for k, v in model.items()}
    m2 = {k: torch.zeros_like(v) for k, v in model.items()}
    gen = torch.Generator(device=dev).manual_seed(0)
    out = {}
    for pattern in os.environ.get("PROBE_PATTERNS", "contiguous,random,groups4,groups16").split(","):
        if pattern == "contiguous":
            index = torch.arange(int(P * frac), device=dev)
        elif pattern == "random":
            index = torch.nonzero(torch.rand(P, device=dev, generator=gen) < frac).reshape(-1)
        else:
            g = int(pattern[6:])
            keep = torch.rand(P // g, device=dev, generator=gen) < frac
            index = torch.nonzero(keep.repeat_interleave(g)).reshape(-1)
        m = int(index.numel())
        flag = torch.ones(m, dtype=torch.bool, device=dev)
        param = {k: torch.rand((m,) + w, device=dev, generator=gen) for k, w in widths.items()}
        grad = {k: torch.rand((m,) + w, device=dev, generator=gen) for k, w in widths.items()}
        entries = [(model[k], param[k], grad[k], m1[k], m2[k], None, 1e-3) for k in widths]
        call = lambda: R._backend.sparse_adam(index, flag, entries, 0.9, 0.999, 0.5, 1e-15)
        call(); call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        out[pattern] = {"rows": m, "ms": round(ms, 3), "GBs": round(m * 59 * 28 / ms / 1e6, 1)}
        del param, grad
    out["env"] = {k: v for k, v in os.environ.items() if k.startswith(("LOGRAST_", "PROBE_"))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

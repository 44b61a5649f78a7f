"""Diagnostic sweep for gpurun: prints/dumps mismatch statistics for every stage instead of asserting,
so one GPU call tells the whole story.  python tests/gpu_diag.py [out.json]"""
import json
import os
import sys
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    import torch
    from oracle import oracle
    from log_amd import rasterizer as R
    import gpu_util as G
    from test_gpu_parity import CASES, _case
    from util import rel_l2
    out = {"device": torch.cuda.get_device_name(0)}
    for name in CASES:
        for flavour in (R.WODILATE, R.UPSTREAM):
            key = f"{name}/{flavour.name.split('_')[-1]}"
            try:
                cam, sc = _case(name)
                bg = (0.3, 0.6, 0.9)
                hf = G.hip_forward(cam, sc, bg, flavour)
                v, of = G.oracle_forward(oracle, cam, sc, bg, flavour)
                st = G.compare_forward(hf, of)
                if flavour.extras:
                    dL = np.random.default_rng(1).random(of["image"].shape, dtype=np.float32)
                    hg = G.hip_backward(hf, dL)
                    og = oracle.backward(v, of, dL)
                    st["grad_rel_l2"] = {k: rel_l2(hg[k], og[k]) for k in og if k in hg}
                    hp = G.hip_project_backward(hf, og["means2D"], og["conic"])
                    st["a6b_isolated_rel_l2"] = {k: rel_l2(hp[k], og[k]) for k in hp}
                out[key] = st
            except Exception:
                out[key] = {"error": traceback.format_exc()}
            print(key, json.dumps(out[key]), flush=True)
    path = sys.argv[1] if len(sys.argv) > 1 else None
    if path:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

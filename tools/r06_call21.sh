#!/bin/bash
# round 6, call 21: the forward's per-Gaussian outputs committed once per chunk: parity, then the bench step
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "forward_bit_exact or backward_vs_oracle or scratch or lazily" > gpurun_out/r06_parity6.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r06_parity6.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin-mode --no-secondary > gpurun_out/b_cc.log 2>/dev/null
python - <<P
import json
d=json.loads(open("gpurun_out/b_cc.log").read().strip().splitlines()[-1])
c,r=d["config"],d["roofline"]
print("opaque %.3f rand %.3f trained %.3f fwd-only %.3f" % (c["ms_per_view"], c["ms_per_view_opacity_rand"], c["ms_per_view_trained_like"], c["forward_only_ms_per_view"]), "fwd %.0f bwd %.0f" % (r["us_blend_fwd"], r["us_blend_bwd"]))
P
done

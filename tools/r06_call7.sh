#!/bin/bash
# round 6, call 7: the restructured prefetch pipelines of the row-split kernels: parity tests, then the bench step with / without hit masks
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "hit_masks or lazily or backward_vs_oracle or forward_bit_exact or empty" > gpurun_out/r06_parity4.log 2>&1; echo "parity rc=$?"; tail -5 gpurun_out/r06_parity4.log
for m in 0 1; do
  LOGRAST_HIT_MASKS=$m python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-forward-only --no-dropin-mode > gpurun_out/b_masks$m.log 2>/dev/null
  python - <<P
import json
d=json.loads(open("gpurun_out/b_masks$m.log").read().strip().splitlines()[-1])
c,r=d["config"],d["roofline"]
print("masks=$m", "opaque %.3f rand %.3f trained %.3f" % (c["ms_per_view"], c["ms_per_view_opacity_rand"], c["ms_per_view_trained_like"]), "fwd %.0f bwd %.0f pbwd %.0f" % (r["us_blend_fwd"], r["us_blend_bwd"], r["us_project_bwd"]))
P
done | tee gpurun_out/r06_pipeline_bench_ab.txt

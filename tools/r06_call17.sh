#!/bin/bash
# round 6, call 17: same-box A/B of two builds of the library (LOGRAST_LIB): prev = the last commit, default = the working tree
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for lib in prev cur prev cur; do
  if [ $lib = prev ]; then export LOGRAST_LIB=$PWD/log_amd/lib/liblograst_prev.so; else unset LOGRAST_LIB; fi
  python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-forward-only --no-dropin-mode > gpurun_out/b_ab.log 2>/dev/null
  python - <<P
import json
d=json.loads(open("gpurun_out/b_ab.log").read().strip().splitlines()[-1])
c,r=d["config"],d["roofline"]
print("$lib", "opaque %.3f rand %.3f trained %.3f" % (c["ms_per_view"], c["ms_per_view_opacity_rand"], c["ms_per_view_trained_like"]), "fwd %.0f bwd %.0f" % (r["us_blend_fwd"], r["us_blend_bwd"]))
P
done | tee gpurun_out/r06_lib_ab.txt

#!/bin/bash
# round 6, call 22: the chain rule's running-sum rows stored as complete lines: tests with gradient sinks, then same-box A/B
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin_modes.py tests/test_gpu_graphs.py -q -m gpu -x -k "fused or accumul or sink or graph or contract or rows" > gpurun_out/r06_sink.log 2>&1; echo "sink tests rc=$?"; tail -3 gpurun_out/r06_sink.log
timeout 900 python -m pytest tests/test_gpu_scale.py -q -m gpu -x -k "c5_band or image_split or band_projection" >> gpurun_out/r06_sink.log 2>&1; echo "band tests rc=$?"; tail -3 gpurun_out/r06_sink.log
for lib in prev cur prev cur; do
  if [ $lib = prev ]; then export LOGRAST_LIB=$PWD/log_amd/lib/liblograst_prev.so; else unset LOGRAST_LIB; fi
  python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-forward-only --no-dropin-mode > gpurun_out/b_ab.log 2>/dev/null
  python - <<P
import json
d=json.loads(open("gpurun_out/b_ab.log").read().strip().splitlines()[-1])
c,r=d["config"],d["roofline"]
print("$lib", "opaque %.3f rand %.3f trained %.3f" % (c["ms_per_view"], c["ms_per_view_opacity_rand"], c["ms_per_view_trained_like"]), "pbwd %.0f" % (r["us_project_bwd"]))
P
done | tee gpurun_out/r06_pbwd_ab.txt

#!/bin/bash
# round 6, call 15: views in flight per GPU at the 30 M headline (one HIP stream each, one graph per (stream, camera))
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for s in 1 2 3 4; do
  python bench.py --steps 10 --warmup 3 --streams $s --no-secondary --no-cpu-baseline --no-forward-only --no-dropin-mode --no-kernel-timing --full-out gpurun_out/b_streams$s.json > gpurun_out/b_streams$s.log 2>/dev/null
  python - <<P
import json
d=json.loads(open("gpurun_out/b_streams$s.log").read().strip().splitlines()[-1])
c=d["config"]
print("streams=$s", "opaque %.3f rand %.3f trained %.3f" % (c["ms_per_view"], c.get("ms_per_view_opacity_rand", 0), c.get("ms_per_view_trained_like", 0)))
P
done | tee gpurun_out/r06_streams_ab.txt

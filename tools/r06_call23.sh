#!/bin/bash
# round 6, call 23: views in flight at C2 (1 M Gaussians)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for s in 2 3 4 6 8; do
  python bench.py --gaussians 1000000 --steps 40 --warmup 5 --streams $s --no-secondary --no-cpu-baseline --no-forward-only --no-dropin-mode --no-kernel-timing --no-rand-variant --no-trained-like > gpurun_out/b_c2s$s.log 2>/dev/null
  python - <<P
import json
d=json.loads(open("gpurun_out/b_c2s$s.log").read().strip().splitlines()[-1])
print("C2 streams=$s ms/view %.4f" % d["config"]["ms_per_view"])
P
done | tee gpurun_out/r06_c2_streams.txt

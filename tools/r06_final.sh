#!/bin/bash
# round 6: what the driver runs at round end (smoke, the bench with its flags), the outputs kept for profiles/
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 --full-out gpurun_out/b_default_full.json > gpurun_out/b_default.log 2> gpurun_out/b_default.err; echo "bench rc=$?"
echo "bench wall seconds: $(( $(date +%s) - t0 ))"
tail -1 gpurun_out/b_default.log | wc -c
tail -1 gpurun_out/b_default.log

#!/bin/bash
# round 6, call 35: the default bench with the one-rank RCCL leg
mkdir -p gpurun_out
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 --full-out gpurun_out/b_default_full.json > gpurun_out/b_default.log 2> gpurun_out/b_default.err; echo "bench rc=$?"
echo "bench wall seconds: $(( $(date +%s) - t0 ))"
echo "stdout lines: $(wc -l < gpurun_out/b_default.log), bytes of the line: $(tail -1 gpurun_out/b_default.log | wc -c)"
python - <<P
import json
l=json.loads(open("gpurun_out/b_default.log").read().strip().splitlines()[-1])
print(l["value"], l["ms_per_step"], l["config"].get("rccl_one_rank_step_ms"), l["config"].get("rccl_one_rank_exchange"))
r=json.load(open("gpurun_out/b_default_full.json"))
print(json.dumps(r["multi_gpu_step_one_rank_rccl"])[:1200])
P

#!/bin/bash
# round 6, after the exchange work: the whole GPU suite, smoke, and the driver's bench command (stdout must be ONE line)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
t0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"
echo "suite wall seconds: $(( $(date +%s) - t0 ))"
tail -3 gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 --full-out gpurun_out/b_default_full.json > gpurun_out/b_default.log 2> gpurun_out/b_default.err; echo "bench rc=$?"
echo "bench wall seconds: $(( $(date +%s) - t0 ))"
echo "stdout lines: $(wc -l < gpurun_out/b_default.log), bytes of the line: $(tail -1 gpurun_out/b_default.log | wc -c)"
tail -1 gpurun_out/b_default.log | cut -c1-700

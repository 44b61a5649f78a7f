#!/bin/bash
# round 6, call 24: randomised parity sweep (tools/fuzz_parity.py) with the final kernels
mkdir -p gpurun_out
for s in 1 2 3; do
timeout 1200 python tools/fuzz_parity.py --cases 400 --seed $s --seconds 900 --out gpurun_out/fuzz_parity_s$s.jsonl > gpurun_out/fuzz_parity_s$s.log 2>&1
echo seed $s rc=$?
grep -E "^FAIL|fuzz_parity:" gpurun_out/fuzz_parity_s$s.log | cut -c1-1500 | head -8
done

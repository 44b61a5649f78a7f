#!/bin/bash
# round 6, call 32: property sweep (bands, bucket accumulation) of tools/fuzz_parity.py
mkdir -p gpurun_out
timeout 1500 python tools/fuzz_parity.py --mode props --cases 300 --seed 4 --seconds 1200 --out gpurun_out/fuzz_props_s4.jsonl > gpurun_out/fuzz_props_s4.log 2>&1
echo rc=$?
grep -E "^FAIL|fuzz_parity:" gpurun_out/fuzz_props_s4.log | cut -c1-900 | head -12

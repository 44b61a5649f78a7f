#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python tools/mask_stats.py 2>gpurun_out/r06_mask_stats.err | tee gpurun_out/r06_mask_stats.jsonl; tail -5 gpurun_out/r06_mask_stats.err
python tools/mask_stats.py --opacity -1 2>/dev/null | tee -a gpurun_out/r06_mask_stats.jsonl
python tools/mask_stats.py --scene trained 2>/dev/null | tee -a gpurun_out/r06_mask_stats.jsonl

#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_getall.py -q -m gpu -x > gpurun_out/r06_train_ops.log 2>&1; echo "train ops rc=$?"; tail -12 gpurun_out/r06_train_ops.log
timeout 900 python tools/bench_log_step.py 40000 7 3 4 2>gpurun_out/c3.err | tee gpurun_out/c3_step.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C3 ms/view', d['ms_per_view'], 'fused step', d.get('ms_per_view_fused_step'), 'bit identical', d.get('fused_step_model_bit_identical'))
print('stages', d['stages_ms']); print('stages fused', d.get('stages_ms_fused_step'))
print('kernels fused', {k:v for k,v in d.get('kernels_us_per_view_fused_step',{}).items() if k in ('sparse_adam','activate_bwd','gather_activate')})
"; tail -3 gpurun_out/c3.err

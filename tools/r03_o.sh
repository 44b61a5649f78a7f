#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
D=$PWD/gpurun_out/r03o
mkdir -p "$D"
LOGRAST_ADAM_ADJACENT=1 timeout 600 python -m pytest tests/test_gpu_train_ops.py -q -m gpu > $D/pytest.log 2>&1; tail -3 $D/pytest.log
LOGRAST_ADAM_ADJACENT=1 timeout 600 python tools/adam_probe.py > $D/adam_adj.json 2> $D/adam.err; tail -1 $D/adam_adj.json; tail -3 $D/adam.err

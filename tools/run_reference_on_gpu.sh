#!/bin/bash
# Runs tests/test_gpu_log_plumbing.py -- the reference's UNMODIFIED classes on the MI355X -- through ONE gpurun call.
# The GPU box has no /root/reference, so a git-ignored copy of the reference's Python package is staged under
# .reference_mount/ for the snapshot and removed again afterwards (never committed: .gitignore).  Usage (in the container):
#   bash tools/run_reference_on_gpu.sh [extra shell commands to run on the box after the test]
# The test's log comes back as gpurun_out/log_plumbing_gpu.log; copy it to profiles/rNN_log_plumbing_gpu.log.
cd "$(dirname "$0")/.." || exit 1
REF=${LOG_REFERENCE:-/root/reference}
[ -d "$REF/LoG" ] || { echo "no reference tree at $REF"; exit 1; }
rm -rf .reference_mount && mkdir -p .reference_mount
cp -r "$REF/LoG" .reference_mount/LoG
find .reference_mount -name '__pycache__' -type d -prune -exec rm -rf {} +
find .reference_mount -name '*.cu' -delete      # (Python only: the CUDA source is not needed for this test)
EXTRA=${1:-true}
/usr/local/graft/bin/gpurun --timeout ${GPURUN_TIMEOUT:-900} -- "${GPURUN_CMD:-mkdir -p gpurun_out; export LOG_REFERENCE=\$PWD/.reference_mount; timeout 600 python -m pytest tests/test_gpu_log_plumbing.py -x -q -s -m gpu > gpurun_out/log_plumbing_gpu.log 2>&1; echo plumbing rc=\$?; tail -15 gpurun_out/log_plumbing_gpu.log; $EXTRA}"
rc=$?
rm -rf .reference_mount
exit $rc

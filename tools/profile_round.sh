#!/bin/bash
# Everything profiles/ is built from, in one gpurun call (run ON the GPU box from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/profile_round.sh r01'
# then, back in the container:  python tools/collect_profiles.py r01
# rocprofv3 rules of this pool: --pmc passes carry --kernel-trace only (no sys/hip/hsa trace domains), one
# counter group per pass (HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes).
TAG=${1:-r01}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
P="python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-kernel-timing"
(timeout 900 python -m pytest tests -m gpu -q --timeout 900 > $D/pytest.log 2>&1; echo pytest_exit=$? >> $D/pytest.log)
timeout 600 python bench.py > $D/b_default.log 2>&1
timeout 600 python bench.py --streams 1 --no-cpu-baseline > $D/b_s1.log 2>&1
timeout 600 python bench.py --gaussians 10000000 --steps 3 --warmup 1 --no-cpu-baseline > $D/b_10M.log 2>&1
timeout 600 python bench.py --gaussians 30000000 --steps 3 --warmup 1 --no-cpu-baseline > $D/b_30M.log 2>&1
rm -rf $D/${TAG}_trace $D/${TAG}_trace_default $D/${TAG}_pmc_fetch $D/${TAG}_pmc_write $D/${TAG}_pmc_sq
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/${TAG}_trace -o c2 -- $P > $D/${TAG}_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/${TAG}_trace_default -o c2 -- \
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $D/${TAG}_trace_default.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/${TAG}_pmc_fetch -o c2 -- $P > $D/${TAG}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/${TAG}_pmc_write -o c2 -- $P > $D/${TAG}_pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU \
  SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $D/${TAG}_pmc_sq -o c2 -- $P > $D/${TAG}_pmc_sq.log 2>&1
timeout 300 python tools/bench_knn.py > $D/knn_bench.log 2>&1
timeout 300 python tools/bench_radius.py 10000000 > $D/radius_10M.log 2>&1
timeout 300 python tools/bench_sh.py 10000000 3 8 > $D/sh_10M.log 2>&1
timeout 300 python tools/bench_lod.py > $D/lod_bench.log 2>&1
timeout 300 python tools/bench_train_ops.py > $D/train_ops_bench.log 2>&1
timeout 300 python tools/bench_get_all.py 1000000 3 > $D/get_all_deg3.log 2>&1
timeout 300 python tools/bench_get_all.py 1000000 1 > $D/get_all_deg1.log 2>&1
timeout 400 python tools/bench_log_step.py > $D/log_step.log 2>&1
rm -rf $D/${TAG}_trace_log_step
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $D/${TAG}_trace_log_step -o step -- \
  python tools/bench_log_step.py 40000 7 1 4 > $D/${TAG}_trace_log_step.log 2>&1
tail -n 3 $D/pytest.log
grep -h '^{' $D/b_default.log | cut -c1-400

#!/bin/bash
# round 6, call 12: (a) what ANY staging of the fill's key stores could gain: the fill without its key stores (experiment
# build, LOGRAST_FILL_ABLATE=2; nothing is sorted or composited: LOGRAST_STOP_AFTER_FILL=1), headline / rand / trained-like;
# (b) SQ counters of the rand and the trained-like view (what bounds their compositing)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
python -m log_amd.build exp -DLR_EXPERIMENTS > /dev/null 2>&1
P="python tools/kernel_probe.py --views 4 --reps 3 --fwd-only --lib log_amd/lib/liblograst_exp.so"
for scene in random trained; do for ab in 0 2; do
  $P --scene $scene --env LOGRAST_STOP_AFTER_FILL=1 LOGRAST_FILL_ABLATE=$ab --tag "fill_${scene}_ablate$ab"
done; done 2>/dev/null | tee $D/r06_fill_bound.jsonl
B="python bench.py --views 4 --steps 1 --warmup 0 --streams 1 --no-graphs --no-cpu-baseline --no-kernel-timing --no-secondary --no-dropin-mode --no-rand-variant --no-forward-only --no-trained-like"
for v in "rand --opacity -1" "trained --scene trained"; do
  set -- $v; name=$1; shift
  rm -rf $D/r06_sq_$name
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU \
    --output-format csv -d $D/r06_sq_$name -o h30 -- $B "$@" > $D/r06_sq_$name.log 2>&1
  echo "## $name"; python tools/pmc_summary.py $(find $D/r06_sq_$name -name '*counter_collection.csv')
done | tee $D/r06_pmc_rand_trained.md
find $D -name '*.csv' -path '*r06_sq_*' -size +20M -delete

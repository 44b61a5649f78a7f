#!/bin/bash
# round 6, call 6: SQ counters of the compositing kernels with and without the forward's hit masks (30 M headline view)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
P="python bench.py --views 4 --steps 1 --warmup 0 --streams 1 --no-graphs --no-cpu-baseline --no-kernel-timing --no-secondary --no-dropin-mode --no-rand-variant --no-forward-only --no-trained-like"
for m in 0 1; do
  rm -rf $D/r06_sq_masks$m
  LOGRAST_HIT_MASKS=$m timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU \
    SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $D/r06_sq_masks$m -o h30 -- $P > $D/r06_sq_masks$m.log 2>&1
  echo "masks=$m"; python tools/pmc_summary.py $(find $D/r06_sq_masks$m -name '*counter_collection.csv') | grep -i "blend\|kernel |\|---"
  rm -rf $D/r06_mix_masks$m
  LOGRAST_HIT_MASKS=$m timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --output-format csv -d $D/r06_mix_masks$m -o h30 -- $P > $D/r06_mix_masks$m.log 2>&1
  python tools/pmc_summary.py $(find $D/r06_mix_masks$m -name '*counter_collection.csv') | grep -i "blend\|kernel |\|---"
done 2>&1 | tee $D/r06_pmc_masks.md
find $D -name '*.csv' -path '*r06_*masks*' -size +20M -delete

#!/bin/bash
# Everything profiles/ is built from, in one gpurun call (run ON the GPU box from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/profile_round.sh r02'
# then, back in the container:  python tools/collect_profiles.py r02
# rocprofv3 rules of this pool: --pmc passes carry --kernel-trace only (no sys/hip/hsa trace domains), one
# counter group per pass (HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes).
# The profiled workload is the bench headline: 30 M Gaussians @1080p (past the 256 MiB Infinity Cache, so
# FETCH_SIZE / WRITE_SIZE are memory-side traffic), views launched eagerly on one stream (rocprofv3 serialises kernels).
TAG=${1:-r06}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
P="python bench.py --views 4 --steps 1 --warmup 0 --streams 1 --no-graphs --no-cpu-baseline --no-kernel-timing --no-secondary --no-dropin-mode --no-rand-variant --no-forward-only --no-trained-like"
# (the LAST stdout line of bench.py is the compact contract object; the full result goes to --full-out)
timeout 900 python bench.py --steps 20 --warmup 5 --full-out $D/b_default_full.json > $D/b_default.log 2>&1
timeout 600 python bench.py --gaussians 10000000 --no-cpu-baseline --no-secondary --no-forward-only --full-out $D/b_10M_full.json > $D/b_10M.log 2>&1
rm -rf $D/${TAG}_trace $D/${TAG}_pmc_fetch $D/${TAG}_pmc_write $D/${TAG}_pmc_sq
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/${TAG}_trace -o h30 -- $P > $D/${TAG}_trace.log 2>&1
# the same step on the trained-like scene (log-normal scales: rects of 5..16 tiles, a few larger): kernel times only
rm -rf $D/${TAG}_trace_trained
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/${TAG}_trace_trained -o t30 -- $P --scene trained > $D/${TAG}_trace_trained.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/${TAG}_pmc_fetch -o h30 -- $P > $D/${TAG}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/${TAG}_pmc_write -o h30 -- $P > $D/${TAG}_pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU \
  SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $D/${TAG}_pmc_sq -o h30 -- $P > $D/${TAG}_pmc_sq.log 2>&1
# instruction mix (round-3 verdict, next #2) and L2 request counts (the fill kernel's bound: requests, not bytes); counter
# names that this rocprofv3 does not know make their pass fail without touching the others
rm -rf $D/${TAG}_pmc_mix $D/${TAG}_pmc_mix2 $D/${TAG}_pmc_tcc
# (at most four counters per pass: nine in one pass were refused -- "Request exceeds the capabilities of the hardware" --
# and the aborted run then sat until its timeout; `timeout -k` + a short limit so that a refused pass costs seconds)
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 \
  --output-format csv -d $D/${TAG}_pmc_mix -o h30 -- $P > $D/${TAG}_pmc_mix.log 2>&1
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR \
  --output-format csv -d $D/${TAG}_pmc_mix2 -o h30 -- $P > $D/${TAG}_pmc_mix2.log 2>&1
timeout -k 5 150 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $D/${TAG}_pmc_tcc -o h30 -- $P > $D/${TAG}_pmc_tcc.log 2>&1
tail -n 3 $D/${TAG}_pmc_mix.log $D/${TAG}_pmc_mix2.log $D/${TAG}_pmc_tcc.log
grep -h '^{' $D/b_default.log | cut -c1-300
ls $D/${TAG}_trace $D/${TAG}_pmc_fetch $D/${TAG}_pmc_write $D/${TAG}_pmc_sq

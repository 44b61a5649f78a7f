#!/bin/bash
# round-2 baseline at the north-star point (30 M Gaussians @1080p): per-kernel HIP-event times + past-L3 PMC traffic
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
B="python bench.py --gaussians 30000000 --views 2 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-kernel-timing"
timeout 400 python bench.py --gaussians 30000000 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > $D/r02_base_30M_s1.log 2>&1
rm -rf $D/r02_pmc_fetch_30M $D/r02_pmc_write_30M $D/r02_pmc_req_30M
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/r02_pmc_fetch_30M -o b30 -- $B > $D/r02_pmc_fetch_30M.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/r02_pmc_write_30M -o b30 -- $B > $D/r02_pmc_write_30M.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $D/r02_pmc_req_30M -o b30 -- $B > $D/r02_pmc_req_30M.log 2>&1
rocprofv3 -L 2>/dev/null | grep -iE "TCC_EA0_(WR|RD)REQ|TCC_HIT|TCC_MISS|WRITE_SIZE|FETCH_SIZE" | head -40 > $D/r02_counters_list.txt
grep -h '^{' $D/r02_base_30M_s1.log | cut -c1-600
ls $D/r02_pmc_fetch_30M $D/r02_pmc_write_30M $D/r02_pmc_req_30M

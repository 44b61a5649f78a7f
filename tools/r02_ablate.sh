#!/bin/bash
# usage: tools/r02_ablate.sh TAG "ENV1=.. ENV2=.." "ENV.." ...   -> per-kernel times at 30 M (single stream) per setting
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out; mkdir -p "$D"
TAG=$1; shift
N=${ABL_N:-30000000}
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs timeout 400 python bench.py --gaussians $N --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary --no-dropin-mode > $D/${TAG}_$i.log 2>&1
  grep -h '^{' $D/${TAG}_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$envs]', 'ms/view', round(d['ms_per_view'],3), ' '.join('%s=%.0f'%(k,v['avg_us']) for k,v in d['kernels'].items()))" || tail -n 5 $D/${TAG}_$i.log
done
if [ -n "$ABL_TRACE" ]; then
  rm -rf $D/${TAG}_trace
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D/${TAG}_trace -o t -- python bench.py --gaussians $N --views 4 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-kernel-timing --no-secondary --no-dropin-mode > $D/${TAG}_trace.log 2>&1
  python - <<PY
import csv
rows=list(csv.DictReader(open("$D/${TAG}_trace/t_kernel_stats.csv")))
for r in rows[:24]:
    print("%-60s calls %4s avg %9.1f us  total %8.2f ms"%(r["Name"].split("(")[0].replace("void ","")[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
fi

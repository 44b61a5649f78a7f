#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
for v in 0 1; do
  LOGRAST_FILL_ABLATE=$v timeout 400 python bench.py --gaussians 30000000 --steps 2 --warmup 1 --streams 1 --no-graphs --no-cpu-baseline --no-secondary --no-dropin-mode > $D/abl_$v.log 2>&1
  grep -h '^{' $D/abl_$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ablate $v', 'ms/view', round(d['ms_per_view'],3), ' '.join('%s=%.0f'%(k,v['avg_us']) for k,v in d['kernels'].items()))" || tail -n 5 $D/abl_$v.log
done
python - <<'PY'
import torch, time
x = torch.empty(30_000_000*16//4, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for n in (30_000_000*4, 30_000_000*3, 30_000_000*4):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); x[:n].zero_(); e.record(); torch.cuda.synchronize()
    print("zero_", n*4/1e6, "MB", round(s.elapsed_time(e)*1e3), "us")
PY

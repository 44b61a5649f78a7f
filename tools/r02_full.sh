#!/bin/bash
# the whole GPU suite + the default bench line (headline, both modes, C2, C3, CPU baseline)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
D=$PWD/gpurun_out
mkdir -p "$D"
TAG=${1:-full}
(timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $D/${TAG}_pytest.log 2>&1; echo pytest_exit=$? >> $D/${TAG}_pytest.log)
tail -n 6 $D/${TAG}_pytest.log
timeout 900 python bench.py > $D/${TAG}_bench.log 2>&1
grep -h '^{' $D/${TAG}_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('headline ms/view', round(d['ms_per_view'],3), 'value', round(d['value']/1e9,3), 'G/s', 'frac_copy', round(d.get('algorithmic_frac_of_measured_copy',0),3), 'copy', round(d.get('measured_copy_GBs',0)))
print('modes', {k:(round(v['ms_per_view'],3)) for k,v in d['modes'].items()})
print('roofline', d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline'].get('traffic'))
s=d.get('secondary',{})
print('c2', {k:(round(v['ms_per_view'],3) if isinstance(v,dict) and 'ms_per_view' in v else None) for k,v in s.get('c2',{}).get('modes',{}).items()} if 'c2' in s else None)
print('c3', json.dumps(s.get('c3'))[:600])
print('cpu', d.get('cpu_baseline'))
" || tail -n 20 $D/${TAG}_bench.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/${TAG}_smoke.log 2>&1; tail -n 2 $D/${TAG}_smoke.log

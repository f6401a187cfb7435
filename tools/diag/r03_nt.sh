#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "bucket or golden" 2>&1 | tail -2
for i in 1 2; do
for WL in c2 c3w6; do
  ST=10; [ $WL = c3w6 ] && ST=3
  timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps $ST --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$WL ms/step', d['ms_per_step'], 'frac', r['frac'], 'scatter ms', r['avg_launch_ms'], 'count', d.get('roofline_count_pass',{}).get('avg_launch_ms'))"
done
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "batch or config3 or golden or synth or config4" 2>&1 | tail -3
for V in "B131k X=1" "B65k SOHIT_BATCH=65536"; do
  set -- $V
  for WL in c3 c3w6; do
    ST=10; [ $WL = c3w6 ] && ST=2
    env $2 timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps $ST --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 $WL ms/step', d['ms_per_step'], 'first', d.get('ms_first_step'), 'nocache', d.get('ms_per_step_hit_cache_off'), 'rows', d['config']['rows'])"
  done
done

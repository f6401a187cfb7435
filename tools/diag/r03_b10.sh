#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "not config4" 2>&1 | tail -3
for V in "PARTS4 X=1" "PARTS1 SOHIT_EMIT_PARTS=1" "PARTS4 X=1" "PARTS1 SOHIT_EMIT_PARTS=1"; do
  set -- $V
  env $2 timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 c3 ms/step', d['ms_per_step'], 'rows', d['config']['rows'], 'nocache', d.get('ms_per_step_hit_cache_off'))"
done

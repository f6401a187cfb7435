#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or synth or batch or bucket" 2>&1 | tail -3
for V in "UG2 X=1" "UG1 SOHIT_UG2=0" "UG1pad SOHIT_UG2=0 SOHIT_UG_LDSPAD=4096" "UG2w2 SOHIT_UG_WAIT=2" ; do
  set -- $V
  for WL in c2 c3w6; do
    ST=10; [ $WL = c3w6 ] && ST=2
    env $2 $3 timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps $ST --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 $WL ms/step', d['ms_per_step'], 'rows', d['config']['rows'], 'ungap', d['stage_ms_per_step'].get('group.ungap'))"
  done
done

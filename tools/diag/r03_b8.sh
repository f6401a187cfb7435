#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "not config4" 2>&1 | tail -4
for V in "PRIV1 X=1" "PRIV0 SOHIT_ALIGN_PRIV=0" "PRIV1 X=1" "PRIV0 SOHIT_ALIGN_PRIV=0"; do
  set -- $V
  env $2 timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 c3 ms/step', d['ms_per_step'], 'rows', d['config']['rows'], 'trace', d['stage_ms_per_step'].get('phase2.trace_pass'), 'rounds', d['stage_ms_per_step'].get('phase2.align_rounds'))"
done
SOHIT_ALIGN_PK=0 timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('PK0+PRIV c3 ms/step', d['ms_per_step'], 'rows', d['config']['rows'], 'trace', d['stage_ms_per_step'].get('phase2.trace_pass'), 'rounds', d['stage_ms_per_step'].get('phase2.align_rounds'))"

#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "not config4" 2>&1 | tail -3
for V in "NEW X=1" "NEW SOHIT_SPEC=0" "NEW X=1"; do
  set -- $V
  env $2 timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['stage_ms_per_step']; print('$1 $2 c3 ms/step', d['ms_per_step'], 'rows', d['config']['rows'], 'rounds', t.get('phase2.align_rounds'), 'trace', t.get('phase2.trace_pass'))"
done
bash tools/diag/profile_bench.sh r03_e_c3only statsonly --no-aux 2>&1 | tail -1

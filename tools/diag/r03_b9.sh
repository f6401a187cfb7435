#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "not config4" 2>&1 | tail -3
for i in 1 2; do
  timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c3 ms/step', d['ms_per_step'], 'rows', d['config']['rows'], 'trace', d['stage_ms_per_step'].get('phase2.trace_pass'), 'rounds', d['stage_ms_per_step'].get('phase2.align_rounds'))"
done
bash tools/diag/profile_bench.sh r03_c_c3only statsonly --no-aux 2>&1 | tail -1

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_b5
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "splits" 2>&1 | tail -40
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -k "not config4 and not splits" 2>&1 | tail -5
for V in "LDS1 X=1" "LDS0 SOHIT_CAND_LDS=0" "LDS1 X=1" "LDS0 SOHIT_CAND_LDS=0"; do
  set -- $V
  env $2 timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 c3 ms/step', d['ms_per_step'], 'first', d.get('ms_first_step'), 'nocache', d.get('ms_per_step_hit_cache_off'), 'rows', d['config']['rows'], d['stage_ms_per_step'].get('group.best_order'))"
done

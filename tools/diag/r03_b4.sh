#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_b4
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "batch or splits or golden" 2>&1 | tail -3
for V in "B131k X=1" "B65k SOHIT_BATCH=65536" "B131k X=1"; do
  set -- $V
  env $2 timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 c3 ms/step', d['ms_per_step'], 'first', d.get('ms_first_step'), 'nocache', d.get('ms_per_step_hit_cache_off'), 'rows', d['config']['rows'])"
done
for V in "B131k X=1" "B16k SOHIT_BATCH=16384" "B3758 SOHIT_BATCH=3758"; do
  set -- $V
  env $2 REPS=1 timeout 1500 python tools/diag/run_config.py 1000000 111111 500000 500032 > gpurun_out/r03_b4/c4_$1.txt 2>&1; echo $1; tail -4 gpurun_out/r03_b4/c4_$1.txt | cut -c1-600
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "batch or config3 or golden or synth" 2>&1 | tail -3
for WL in c3 c3w6 c2; do
    ST=10; [ $WL = c3w6 ] && ST=2
    timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps $ST --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$WL ms/step', d['ms_per_step'], 'rows', d['config']['rows'])"
done
REPS=1 timeout 1500 python tools/diag/run_config.py 1000000 111111 500000 500032 > gpurun_out/c4_1M_b.txt 2>&1; tail -5 gpurun_out/c4_1M_b.txt | cut -c1-1500

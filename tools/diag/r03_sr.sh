#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "not config4" 2>&1 | tail -3
timeout 900 python tools/diag/fuzz_parity.py 40 5151 2>&1 | tail -1
FUZZ_LONG=1 timeout 900 python tools/diag/fuzz_parity.py 10 78 2>&1 | tail -1
for i in 1 2; do
  timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['stage_ms_per_step']; print('c3 ms/step', d['ms_per_step'], 'rows', d['config']['rows'], 'rounds', t.get('phase2.align_rounds'), 'trace', t.get('phase2.trace_pass'), 'mk', t.get('phase2.mktasks'), d['other_kernels']['k_align_Gcells_per_s'])"
done

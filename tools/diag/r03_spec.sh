#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "speculative or query_ranges or golden" 2>&1 | tail -3
SOHIT_SPEC=1 timeout 900 python tools/diag/fuzz_parity.py 40 991 2>&1 | tail -1
for V in "P2 X=1" "P1 SOHIT_SPEC_PARTS=1" "P4 SOHIT_SPEC_PARTS=4" "P2 X=1" "P1 SOHIT_SPEC_PARTS=1" "P3 SOHIT_SPEC_PARTS=3"; do
  set -- $V
  env $2 timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['stage_ms_per_step']; print('$1 c3 ms/step', d['ms_per_step'], 'rows', d['config']['rows'], 'rounds', t.get('phase2.align_rounds'), 'trace', t.get('phase2.trace_pass'))"
done

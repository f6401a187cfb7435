#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_b11
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "query_ranges or splits" 2>&1 | tail -3
timeout 900 python tools/diag/fuzz_parity.py 60 4242 2>&1 | tail -2
FUZZ_LONG=1 timeout 900 python tools/diag/fuzz_parity.py 12 77 2>&1 | tail -2
for V in "P8 SOHIT_EMIT_PARTS=8" "P4 X=1" "P8 SOHIT_EMIT_PARTS=8" "P4 X=1"; do
  set -- $V
  env $2 timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 c3 ms/step', d['ms_per_step'], 'rows', d['config']['rows'])"
done
timeout 600 python bench.py --workload c3w6 --no-cpu-baseline --no-aux --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c3w6 ms/step', d['ms_per_step'], 'rows', d['config']['rows'])"
timeout 600 python bench.py --workload c2 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c2 ms/step', d['ms_per_step'], 'rows', d['config']['rows'])"

#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "not config4" 2>&1 | tail -4
for i in 1 2; do
timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c3 ms/step', d['ms_per_step'], 'first', d.get('ms_first_step'), 'nocache', d.get('ms_per_step_hit_cache_off'), 'rows', d['config']['rows'], d['other_kernels']['k_align_ms_per_step'], d['stage_ms_per_step'])"
done
SOHIT_ALIGN_PK=0 timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('PK0 c3 ms/step', d['ms_per_step'], 'rows', d['config']['rows'])"

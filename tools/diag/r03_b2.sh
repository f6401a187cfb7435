#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for V in "EQ0 SOHIT_BATCH_EQUAL=0" "EQ1 SOHIT_BATCH_EQUAL=1" "B33k SOHIT_BATCH=33334" "B25k SOHIT_BATCH=25000" "B100k SOHIT_BATCH=100000"; do
    set -- $V
    env $2 timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 c3 ms/step', d['ms_per_step'], 'rows', d['config']['rows'])"
done
done

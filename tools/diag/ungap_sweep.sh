#!/bin/bash
# k_ungap tuning sweep on BASELINE config 2 (GPU box): chunks per loop iteration x bookkeeping threshold.
cd "$(dirname "$0")/../.."
run() { env "$@" python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['stage_ms_per_step']['group.ungap'], d['ms_per_step'])"; }
for cpi in ${CPIS:-1 2 3}; do for w in ${WAITS:-16 20 24}; do run SOHIT_UG_CPI=$cpi SOHIT_UG_WAIT=$w; done; done

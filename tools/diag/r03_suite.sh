#!/bin/bash
cd $GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests -q -m gpu ) 2>&1 | tail -8
for A in 200 600; do ( SOHIT_BUCKET_MIN=0 SOHIT_BUCKET_AVG=$A timeout 600 python tools/diag/fuzz_parity.py 25 $((13000 + A)) ) > gpurun_out/fuzz_avg$A.log 2>&1; echo "bucket avg $A: $(grep -c ' ok ' gpurun_out/fuzz_avg$A.log) ok"; grep -v " ok " gpurun_out/fuzz_avg$A.log | tail -2; done
SOHIT_BATCH=50000 timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('batch 50000: c3 ms/step', d['ms_per_step'], 'rows', d['config']['rows'])"

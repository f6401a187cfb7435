#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_c4
REPS=1 timeout 1500 python tools/diag/run_config.py 1000000 111111 500000 500032 > gpurun_out/r03_c4/c4_1M.txt 2>&1; tail -6 gpurun_out/r03_c4/c4_1M.txt | cut -c1-900
SOHIT_CPU_FULL=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-aux > gpurun_out/r03_c4/cpu_full.json 2>/dev/null; python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03_c4/cpu_full.json") if l.startswith('{')][-1])
print(json.dumps(d["cpu_baseline"])[:900])
PY

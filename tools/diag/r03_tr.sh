#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_tr
timeout 900 python -m pytest tests/test_find_cluster.py -x -q -m gpu 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "not config4_full and not bench and not chain" 2>&1 | tail -5
( timeout 600 python tools/diag/fuzz_parity.py 50 12001 ) > gpurun_out/r03_tr/fuzz.log 2>&1; echo "fuzz: $(grep -c ' ok ' gpurun_out/r03_tr/fuzz.log) ok"; grep -v " ok " gpurun_out/r03_tr/fuzz.log | tail -2
for TP in 1 0; do
  SOHIT_TRACE_PK=$TP timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null > gpurun_out/r03_tr/c3_tp$TP.json
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03_tr/c3_tp$TP.json") if l.startswith('{')][-1])
s=d["stage_ms_per_step"]
print("tracepk=$TP c3 ms/step", d["ms_per_step"], "rows", d["config"]["rows"], "align_rounds", s.get("phase2.align_rounds"), "trace", s.get("phase2.trace_pass"))
PY
done

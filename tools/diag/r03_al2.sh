#!/bin/bash
# round 3: packed aligner with the lane-private score table -- parity subset, fuzz, config-3 timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_al2
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or synth or flag or long or tile or ragged or quirk or config3" 2>&1 | tail -4
( timeout 600 python tools/diag/fuzz_parity.py 40 6101 ) > gpurun_out/r03_al2/fuzz.log 2>&1; echo "fuzz: $(grep -c ' ok ' gpurun_out/r03_al2/fuzz.log) ok"; grep -v " ok " gpurun_out/r03_al2/fuzz.log | tail -2
for PK in 1 0; do
  SOHIT_ALIGN_PK=$PK timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null > gpurun_out/r03_al2/c3_pk$PK.json
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03_al2/c3_pk$PK.json") if l.startswith('{')][-1])
s=d["stage_ms_per_step"]
print("pk=$PK c3 ms/step", d["ms_per_step"], "rows", d["config"]["rows"], "align_rounds", s.get("phase2.align_rounds"), "trace", s.get("phase2.trace_pass"), "Gcells/s", d["other_kernels"]["k_align_Gcells_per_s"])
PY
done

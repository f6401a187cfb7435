#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_ug
bash tools/diag/r03_pmc2.sh k_align_pk r03_pmc_alpk --workload c3 2>&1 | tail -40
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "synth or golden or bucket or family or config4_shape or tandem or homopol or stale" 2>&1 | tail -3
( timeout 600 python tools/diag/fuzz_parity.py 40 8243 ) > gpurun_out/r03_ug/fuzz.log 2>&1; echo "fuzz: $(grep -c ' ok ' gpurun_out/r03_ug/fuzz.log) ok"; grep -v " ok " gpurun_out/r03_ug/fuzz.log | tail -2
for UX in 1 0; do
for WL in c2 c3w6; do
    ST=8; [ $WL = c3w6 ] && ST=2
    SOHIT_UG_X=$UX timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps $ST --warmup 1 2>/dev/null > gpurun_out/r03_ug/${WL}_$UX.json
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03_ug/${WL}_$UX.json") if l.startswith('{')][-1])
s=d["stage_ms_per_step"]
print("ux=$UX $WL ms/step", d["ms_per_step"], "rows", d["config"]["rows"], "ungap", s.get("group.ungap"), "grp", s.get("group.bucket_group"))
PY
done
done

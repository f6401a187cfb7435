#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_cs
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "synth or bucket or family or config4_shape or golden" 2>&1 | tail -3
( SOHIT_BUCKET_MIN=0 timeout 600 python tools/diag/fuzz_parity.py 40 11243 ) > gpurun_out/r03_cs/fuzz_bkt.log 2>&1; echo "bucket forced: $(grep -c ' ok ' gpurun_out/r03_cs/fuzz_bkt.log) ok"; grep -v " ok " gpurun_out/r03_cs/fuzz_bkt.log | tail -2
for CS in 1 0; do
for WL in c3w6 c2; do
    ST=8; [ $WL = c3w6 ] && ST=2
    SOHIT_CAND_SEGSORT=$CS timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps $ST --warmup 1 2>/dev/null > gpurun_out/r03_cs/${WL}_$CS.json
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03_cs/${WL}_$CS.json") if l.startswith('{')][-1])
s=d["stage_ms_per_step"]
print("candseg=$CS $WL ms/step", d["ms_per_step"], "nocache", d.get("ms_per_step_hit_cache_off"), "rows", d["config"]["rows"], "best_order", s.get("group.best_order"), "ungap", s.get("group.ungap"))
PY
done
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_wpe
for W in 6 5 4; do
for WL in c3w6 c2; do
    ST=8; [ $WL = c3w6 ] && ST=2
    SOHIT_BK_WPE=$W timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps $ST --warmup 1 2>/dev/null > gpurun_out/r03_wpe/${WL}_$W.json
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03_wpe/${WL}_$W.json") if l.startswith('{')][-1])
r=d["roofline"]; c=d.get("roofline_count_pass") or {}
print("wpe=$W $WL ms/step", d["ms_per_step"], "scatter ms", r["avg_launch_ms"], "frac", r["frac"], "count ms", c.get("avg_launch_ms"))
PY
done
done
python tools/diag/r03_c5_stages.py 100000 2>&1 | tail -14

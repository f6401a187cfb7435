#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_nt
for NT in 0 1; do
for WL in c3w6 c2; do
    ST=8; [ $WL = c3w6 ] && ST=2
    SOHIT_UG_X=0 SOHIT_BK_NT=$NT timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps $ST --warmup 1 2>/dev/null > gpurun_out/r03_nt/${WL}_$NT.json
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03_nt/${WL}_$NT.json") if l.startswith('{')][-1])
r=d["roofline"]; c=d.get("roofline_count_pass") or {}
s=d["stage_ms_per_step"]
print("nt=$NT $WL ms/step", d["ms_per_step"], "first", d.get("ms_first_step"), "nocache", d.get("ms_per_step_hit_cache_off"), "rows", d["config"]["rows"], "scatter ms", r["avg_launch_ms"], "frac", r["frac"], "count ms", c.get("avg_launch_ms"),
      "stages: count", s.get("seed.bucket_count"), "scatter", s.get("seed.bucket_scatter"), "grp", s.get("group.bucket_group"), "ungap", s.get("group.ungap"))
PY
done
done

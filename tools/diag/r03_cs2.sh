#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_cs2
timeout 900 python -m pytest tests/test_find_cluster.py -x -q -m gpu 2>&1 | tail -3
for CFG in -1 1 2 3; do
    E=""; [ $CFG != -1 ] && E="SOHIT_CSEG_CFG=$CFG"
    env $E timeout 600 python bench.py --workload c3w6 --no-cpu-baseline --no-aux --steps 2 --warmup 1 2>/dev/null > gpurun_out/r03_cs2/c3w6_$CFG.json
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03_cs2/c3w6_$CFG.json") if l.startswith('{')][-1])
s=d["stage_ms_per_step"]
print("csegcfg=$CFG c3w6 ms/step", d["ms_per_step"], "best_order", s.get("group.best_order"))
PY
done

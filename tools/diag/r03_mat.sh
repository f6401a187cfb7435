#!/bin/bash
# round 3: micro-benchmarks; tile-major count matrix parity + timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_mat
./tools/ubench/gather 2>&1 | tee gpurun_out/r03_mat/ubench.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "synth or golden or bucket or family or config4_shape" 2>&1 | tail -3
( SOHIT_POISON=0xFF SOHIT_BUCKET_MIN=0 timeout 600 python tools/diag/fuzz_parity.py 30 7243 ) > gpurun_out/r03_mat/fuzz_bkt.log 2>&1; echo "poison+bucket: $(grep -c ' ok ' gpurun_out/r03_mat/fuzz_bkt.log) ok"; grep -v " ok " gpurun_out/r03_mat/fuzz_bkt.log | tail -2
for WL in c3w6 c2; do
    ST=8; [ $WL = c3w6 ] && ST=2
    timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps $ST --warmup 1 2>/dev/null > gpurun_out/r03_mat/${WL}.json
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03_mat/${WL}.json") if l.startswith('{')][-1])
r=d["roofline"]; c=d.get("roofline_count_pass") or {}
s=d["stage_ms_per_step"]
print("$WL ms/step", d["ms_per_step"], "rows", d["config"]["rows"], "scatter ms", r["avg_launch_ms"], "frac", r["frac"], "count ms", c.get("avg_launch_ms"), "frac", c.get("frac"),
      "stages: count", s.get("seed.bucket_count"), "scatter", s.get("seed.bucket_scatter"), "grp", s.get("group.bucket_group"), "ungap", s.get("group.ungap"))
PY
done

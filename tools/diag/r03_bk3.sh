#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_bk3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "synth or bucket or family or config4_shape" 2>&1 | tail -3
( SOHIT_POISON=0xFF SOHIT_BUCKET_MIN=0 timeout 600 python tools/diag/fuzz_parity.py 30 9243 ) > gpurun_out/r03_bk3/fuzz_bkt.log 2>&1; echo "poison+bucket: $(grep -c ' ok ' gpurun_out/r03_bk3/fuzz_bkt.log) ok"; grep -v " ok " gpurun_out/r03_bk3/fuzz_bkt.log | tail -2
for WL in c3w6 c2; do
    ST=8; [ $WL = c3w6 ] && ST=2
    timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps $ST --warmup 1 2>/dev/null > gpurun_out/r03_bk3/${WL}.json
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03_bk3/${WL}.json") if l.startswith('{')][-1])
r=d["roofline"]; c=d.get("roofline_count_pass") or {}
print("$WL ms/step", d["ms_per_step"], "scatter ms", r["avg_launch_ms"], "frac", r["frac"], "count ms", c.get("avg_launch_ms"))
PY
done
bash tools/diag/r03_pmc2.sh k_bkt_pass r03_pmc_bkt --workload c2 2>&1 | grep -E "k_bkt_pass.*(INSTS_VALU|launches)" | head -4

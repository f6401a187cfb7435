#!/bin/bash
# round 3: scatter-pass variants (LDS-staged coalesced stores, bank-skewed histogram copies) -- parity subset, then timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_bk
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "synth or golden or bucket or family or config4_shape or ragged or quirk" 2>&1 | tail -5
( SOHIT_POISON=0xFF SOHIT_BUCKET_MIN=0 timeout 600 python tools/diag/fuzz_parity.py 40 5243 ) > gpurun_out/r03_bk/fuzz_bkt.log 2>&1; echo "poison+bucket: $(grep -c ' ok ' gpurun_out/r03_bk/fuzz_bkt.log) ok"; grep -v " ok " gpurun_out/r03_bk/fuzz_bkt.log | tail -2
( SOHIT_BUCKET_MIN=0 SOHIT_BUCKET_AVG=48 SOHIT_BK_STAGED=0 timeout 600 python tools/diag/fuzz_parity.py 30 5245 ) > gpurun_out/r03_bk/fuzz_narrow.log 2>&1; echo "narrow buckets unstaged: $(grep -c ' ok ' gpurun_out/r03_bk/fuzz_narrow.log) ok"; grep -v " ok " gpurun_out/r03_bk/fuzz_narrow.log | tail -2
for V in "1 1" "0 1"; do
  set -- $V
  for WL in c3w6 c2; do
    ST=8; [ $WL = c3w6 ] && ST=2
    SOHIT_BK_STAGED=$1 SOHIT_BK_SKEW=$2 timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps $ST --warmup 1 2>/dev/null > gpurun_out/r03_bk/${WL}_$1_$2.json
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03_bk/${WL}_$1_$2.json") if l.startswith('{')][-1])
r=d["roofline"]; c=d.get("roofline_count_pass") or {}
print("staged=$1 skew=$2 $WL ms/step", d["ms_per_step"], "rows", d["config"]["rows"], "scatter ms", r["avg_launch_ms"], "frac", r["frac"], "count ms", c.get("avg_launch_ms"), "frac", c.get("frac"),
      "grp", d["stage_ms_per_step"].get("group.bucket_group"), "ungap", d["stage_ms_per_step"].get("group.ungap"))
PY
  done
done

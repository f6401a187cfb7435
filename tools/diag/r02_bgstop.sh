cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bkt
for S in 0 1 2 3 4; do
  SOHIT_BG_STOP=$S timeout 300 python bench.py --workload c2 --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/bkt/stop$S.json 2> gpurun_out/bkt/stop$S.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/bkt/stop$S.json") if l.startswith("{")][-1])
    print("STOP=$S", d["ms_per_step"], {k:v for k,v in d["stage_ms_per_step"].items() if "bucket" in k or "ungap" in k})
except Exception as e: print("STOP=$S failed", e)
PY
done

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bkt
( timeout 900 python tools/diag/fuzz_parity.py 60 20261003 ) > gpurun_out/bkt/fuzz.log 2>&1; tail -3 gpurun_out/bkt/fuzz.log; grep -c " ok " gpurun_out/bkt/fuzz.log
for B in 1 0; do
  SOHIT_BUCKET=$B timeout 600 python bench.py --workload c3w6 --no-cpu-baseline --no-aux --steps 2 --warmup 1 > gpurun_out/bkt/c3w6_b$B.json 2> gpurun_out/bkt/c3w6_b$B.err
  SOHIT_BUCKET=$B timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 5 --warmup 1 > gpurun_out/bkt/c3_b$B.json 2> gpurun_out/bkt/c3_b$B.err
  python - <<PY
import json
for wl in ("c3w6","c3"):
    d=json.loads([l for l in open("gpurun_out/bkt/%s_b$B.json"%wl) if l.startswith("{")][-1])
    print("BUCKET=$B", wl, "ms_per_step", d["ms_per_step"], "rows", d["config"]["rows"], {k:v for k,v in d["stage_ms_per_step"].items() if v>1.0})
PY
done

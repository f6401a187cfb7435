cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bkt
for B in 16384 32768 65536 131072; do
  SOHIT_BATCH=$B timeout 300 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 5 --warmup 1 > gpurun_out/bkt/batch$B.json 2> gpurun_out/bkt/batch$B.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/bkt/batch$B.json") if l.startswith("{")][-1])
    print("BATCH=$B c3 ms_per_step", d["ms_per_step"], "rows", d["config"]["rows"])
except Exception as e: print("BATCH=$B failed", e, open("gpurun_out/bkt/batch$B.err").read()[-300:])
PY
done

cd $GRAFT_REPO_ROOT
for V in 0 1; do
  if [ $V = 1 ]; then export SOHIT_ALIGN_SORT=1; fi
  timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 6 --warmup 1 > gpurun_out/alsort_$V.json 2> gpurun_out/alsort_$V.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/alsort_$V.json") if l.startswith("{")][-1])
print("sort=$V ms_per_step", d["ms_per_step"], "rows", d["config"]["rows"], {k:v for k,v in d["stage_ms_per_step"].items() if v>2})
PY
done

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bkt
( timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q ) > gpurun_out/bkt/pytest.log 2>&1
tail -4 gpurun_out/bkt/pytest.log
for WL in c2 c3w6 c3; do
  ST=10; [ $WL = c3w6 ] && ST=2
  timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps $ST --warmup 1 > gpurun_out/bkt/$WL.json 2> gpurun_out/bkt/$WL.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bkt/$WL.json") if l.startswith("{")][-1])
print("$WL ms_per_step", d["ms_per_step"], "rows", d["config"]["rows"], {k:v for k,v in d["stage_ms_per_step"].items() if v>0.6})
PY
done

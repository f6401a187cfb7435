cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bkt
( timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q ) > gpurun_out/bkt/pytest.log 2>&1
tail -3 gpurun_out/bkt/pytest.log
for WL in c3 c2; do
  timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps 10 --warmup 2 > gpurun_out/bkt/tb_$WL.json 2> gpurun_out/bkt/tb_$WL.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bkt/tb_$WL.json") if l.startswith("{")][-1])
t=d["stage_ms_per_step"]
print("$WL ms_per_step", d["ms_per_step"], "rows", d["config"]["rows"], {k:t[k] for k in ("phase2.align_rounds","phase2.trace_pass","phase2.csort","phase2.stop") if k in t})
PY
done

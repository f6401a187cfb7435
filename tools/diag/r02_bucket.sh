# bucketed binning: parity first, then A/B timing on config 2
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bkt
( timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q ) > gpurun_out/bkt/pytest.log 2>&1
tail -25 gpurun_out/bkt/pytest.log
for B in 1 0; do
  SOHIT_BUCKET=$B timeout 600 python bench.py --workload c2 --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/bkt/c2_b$B.json 2> gpurun_out/bkt/c2_b$B.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bkt/c2_b$B.json") if l.startswith("{")][-1])
print("BUCKET=$B c2 ms_per_step", d["ms_per_step"], "rows", d["config"]["rows"], {k:v for k,v in d["stage_ms_per_step"].items() if v>0.2}, d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"])
PY
done

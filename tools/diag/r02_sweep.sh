cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bkt
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --workload c3w6 --no-cpu-baseline --no-aux --steps 2 --warmup 1 > gpurun_out/bkt/sw_$tag.json 2> gpurun_out/bkt/sw_$tag.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/bkt/sw_$tag.json") if l.startswith("{")][-1])
    t=d["stage_ms_per_step"]
    print("$tag", d["ms_per_step"], {k:t[k] for k in ("seed.bucket_count","seed.bucket_scatter","group.bucket_group","group.ungap","group.best_order") if k in t})
except Exception as e: print("$tag failed", e)
PY
}
run avg768 SOHIT_BUCKET_AVG=768
run avg1024 SOHIT_BUCKET_AVG=1024
run avg2048 SOHIT_BUCKET_AVG=2048
run avg2560 SOHIT_BUCKET_AVG=2560
run cpi2 SOHIT_UG_CPI=2
run wait32 SOHIT_UG_WAIT=32
run wait12 SOHIT_UG_WAIT=12

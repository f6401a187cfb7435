cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bkt
for WL in c2 c3w6; do
  ST=10; [ $WL = c3w6 ] && ST=2
  timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps $ST --warmup 1 > gpurun_out/bkt/ab_$WL.json 2> gpurun_out/bkt/ab_$WL.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bkt/ab_$WL.json") if l.startswith("{")][-1])
t=d["stage_ms_per_step"]
print("$WL ms_per_step", d["ms_per_step"], "rows", d["config"]["rows"], {k:t[k] for k in ("seed.bucket_count","seed.bucket_scatter","group.bucket_group","group.ungap","group.best_order") if k in t})
PY
done

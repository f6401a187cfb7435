cd $GRAFT_REPO_ROOT
for A in 2048 3072 3584; do
for WL in c2 c3w6; do
  ST=8; [ $WL = c3w6 ] && ST=2
  SOHIT_BUCKET_AVG=$A timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps $ST --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['stage_ms_per_step']; print('avg $A $WL', d['ms_per_step'], {k:s.get(k) for k in ('group.bucket_group','group.ungap','group.best_order','seed.bucket_scatter','seed.bucket_count')})"
done; done

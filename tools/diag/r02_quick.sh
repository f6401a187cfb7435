cd $GRAFT_REPO_ROOT
for WL in c2 c3w6; do
  ST=8; [ $WL = c3w6 ] && ST=2
  timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-aux --steps $ST --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$WL', d['ms_per_step'], d['config']['rows'], d['stage_ms_per_step'].get('group.bucket_group'))"
done

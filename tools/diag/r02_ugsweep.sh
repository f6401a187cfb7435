cd $GRAFT_REPO_ROOT
for W in 20 28 36 48; do
  SOHIT_UG_WAIT=$W timeout 300 python bench.py --workload c2 --no-cpu-baseline --no-aux --steps 8 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('wait $W', d['ms_per_step'], d['stage_ms_per_step'].get('group.ungap'))"
done
for C in 2 3; do
  SOHIT_UG_CPI=$C timeout 300 python bench.py --workload c2 --no-cpu-baseline --no-aux --steps 8 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cpi $C', d['ms_per_step'], d['stage_ms_per_step'].get('group.ungap'))"
done

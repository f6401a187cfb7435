#!/bin/bash
# A/B of one switch on a bench workload, runs interleaved:  bash tools/diag/ab_env.sh SOHIT_TB_EARLY "1 0" 3 --workload c3
# -> gpurun_out/ab_<VAR>_<workload>.txt: ms_per_step, rows and stage times of every run
VAR=$1; VALS=$2; REP=${3:-3}; shift; shift; shift
W=c3; for a in "$@"; do case $a in --workload) ;; *) W=$a;; esac; done
R=$GRAFT_REPO_ROOT; cd $R; OUT=gpurun_out/ab_${VAR}_$W.txt; : > $OUT
for i in $(seq $REP); do
  for v in $VALS; do
    env $VAR=$v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-aux "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', d['ms_per_step'], d['config'].get('rows'), json.dumps(d.get('stage_ms_per_step')))
" >> $OUT
  done
done
cat $OUT

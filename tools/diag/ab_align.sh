#!/bin/bash
# kernel times of the score-only aligner, k_align_lane against k_align_pk, on BASELINE config 3 (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ab_align; mkdir -p $OUT; cd $R
for v in 1 0; do
  export SOHIT_ALIGN_LANE=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v$v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aux > $OUT/v$v.log 2>&1
  find $OUT/v$v -name "*kernel_trace.csv" -delete; find $OUT/v$v -name "*.db" -delete
  f=$(find $OUT/v$v -name "*kernel_stats.csv" | head -1)
  echo "== SOHIT_ALIGN_LANE=$v"; python3 -c "
import csv
for r in list(csv.DictReader(open('$f')))[:6]: print('%-40s %5s %9.1f us'%(r['Name'].split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1e3))
"; grep -o '"ms_per_step": [0-9.]*' $OUT/v$v.log | head -1
done

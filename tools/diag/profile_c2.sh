#!/bin/bash
# rocprofv3 passes over bench.py (BASELINE config 2) on the GPU box; outputs under gpurun_out/prof_<tag>/.
#   1. --kernel-trace --stats           per-kernel durations
#   2. --pmc FETCH_SIZE                 L2 -> fabric read requests   (own pass)
#   3. --pmc WRITE_SIZE                 L2 -> fabric write requests  (own pass)
# Summaries are distilled by tools/diag/summarize_prof.py into profiles/.
TAG=${1:-r01_d}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
if [ "$2" != "statsonly" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
fi
find $OUT -name "*.csv" | head -20
tail -1 $OUT/stats.log | cut -c1-400

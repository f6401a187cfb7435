#!/bin/bash
# rocprofv3 passes over bench.py (default workload: BASELINE config 3 + the weight-6 sub-run); outputs under gpurun_out/prof_<tag>/.
#   1. --kernel-trace --stats           per-kernel durations (only the stats csv is kept: the trace itself is hundreds of MB)
#   2. --pmc FETCH_SIZE / WRITE_SIZE    own passes (tools/diag/summarize_prof.py applies the gfx950 correction)
# usage: bash tools/diag/profile_bench.sh <tag> [statsonly] [extra bench.py args]
TAG=${1:-r02_a}; MODE=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_trace.csv" -delete; find $OUT/stats -name "*.db" -delete
if [ "$MODE" != "statsonly" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/fetch.log 2>&1
find $OUT/fetch -name "*kernel_trace.csv" -delete
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/write.log 2>&1
find $OUT/write -name "*kernel_trace.csv" -delete
fi
du -sh $OUT; find $OUT -name "*.csv" | head -20
tail -1 $OUT/stats.log | cut -c1-600

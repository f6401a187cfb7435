#!/bin/bash
# per-kernel durations of one bench workload: bash tools/diag/kstats.sh <tag> [bench args]   -> gpurun_out/kstats_<tag>.csv (rocprofv3 --kernel-trace --stats)
TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/kstats_$TAG; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux "$@" > $OUT.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
cp $(find $OUT -name "*kernel_stats.csv" | head -1) $R/gpurun_out/kstats_$TAG.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/kstats_$TAG.csv")))
for r in rows[:28]:
    print("%-60s %5s %10.1f us avg %6.2f%%"%(r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY

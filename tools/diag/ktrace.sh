#!/bin/bash
# start time / duration / stream of every launch of the kernels matching a regex in ONE bench step:  bash tools/diag/ktrace.sh <tag> <regex> [bench args]
TAG=$1; RE=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ktrace_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-aux "$@" > $OUT.log 2>&1
python3 - <<PY
import csv,glob,re
f=glob.glob("$OUT/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
# last step only: keep the rows after the last k_layout-free gap? simply print the matches with their neighbours' names
pat=re.compile(r"$RE")
prev_end=None
for i,r in enumerate(rows):
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if pat.search(r["Kernel_Name"]):
        print("%10.3f ms  dur %8.1f us  q %s  grid %s  %s"%((s-t0)/1e6,(e-s)/1e3,r.get("Queue_Id"),r.get("Grid_Size"),r["Kernel_Name"].split("(")[0][-40:]))
PY
find $OUT -name "*kernel_trace.csv" -delete

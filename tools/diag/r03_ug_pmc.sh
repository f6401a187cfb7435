#!/bin/bash
# round 3: L2 request counters of k_ungap (is the extension kernel bound by the L1 miss stream of its per-lane 8-byte gathers?)
K=${1:-k_ungap}; TAG=${2:-r03_ug_pmc}; shift; shift
ARGS=${@:---workload c2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 400 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "$K" --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-aux $ARGS > $OUT/a.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-include-regex "$K" --kernel-trace --output-format csv -d $OUT/b -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-aux $ARGS > $OUT/b.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv,glob,collections
for d in ("a","b"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv"%d):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0][-30:],r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(agg.items()): print(k, "n=%d avg=%.4g"%(len(v),sum(v)/len(v)))
    for l in open("$OUT/%s.log"%d):
        if "rror" in l: print(l.strip()[:200])
PY

#!/bin/bash
# SQ + TCC counter passes for one kernel regex over bench.py:  bash tools/diag/r03_pmc2.sh <regex> <tag> [bench args]
K=${1:-k_align_pk}; TAG=${2:-r03_pmc2}; shift; shift
ARGS=${@:---workload c3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-include-regex "$K" --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-aux $ARGS > $OUT/a.log 2>&1
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --kernel-include-regex "$K" --kernel-trace --output-format csv -d $OUT/b -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-aux $ARGS > $OUT/b.log 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU --kernel-include-regex "$K" --kernel-trace --output-format csv -d $OUT/c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-aux $ARGS > $OUT/c.log 2>&1
timeout 400 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "$K" --kernel-trace --output-format csv -d $OUT/d -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-aux $ARGS > $OUT/d.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("a","b","c","d"):
    dur=collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/*/*_kernel_trace.csv"%d):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"].split("(")[0][-30:]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k,v in sorted(dur.items()): print(d, k, "launches %d avg %.1f us total %.2f ms"%(len(v), sum(v)/len(v)/1e3, sum(v)/1e6))
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv"%d):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0][-30:],r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(agg.items()): print(k, "n=%d avg=%.4g sum=%.4g"%(len(v),sum(v)/len(v),sum(v)))
    for l in open("$OUT/%s.log"%d):
        if "rror" in l: print(l.strip()[:200])
PY
find $OUT -name "*kernel_trace.csv" -delete

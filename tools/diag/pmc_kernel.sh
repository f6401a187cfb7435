#!/bin/bash
# SQ counter passes for one kernel (regex) over bench.py:  bash tools/diag/pmc_kernel.sh k_ungap tag [bench args...]
K=${1:-k_ungap}; TAG=${2:-pmc}; shift; shift
ARGS=${@:---workload c2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-include-regex "$K" --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-aux $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU --kernel-include-regex "$K" --kernel-trace --output-format csv -d $OUT/b -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-aux $ARGS > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-include-regex "$K" --kernel-trace --output-format csv -d $OUT/c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-aux $ARGS > $OUT/c.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv,glob,collections
for d in ("a","b","c"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv"%d):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0][-30:],r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(agg.items()): print(k, "n=%d avg=%.4g"%(len(v),sum(v)/len(v)))
    for l in open("$OUT/%s.log"%d):
        if "rror" in l: print(l.strip()[:200])
PY

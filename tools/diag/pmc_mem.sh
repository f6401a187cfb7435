#!/bin/bash
# memory-side counters for one kernel regex:  bash tools/diag/pmc_mem.sh k_ungap_single tag
K=${1:-k_ungap_single}; TAG=${2:-pmcm}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS --kernel-include-regex "$K" --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/a.log 2>&1
# NOTE: a pass with TA_*_sum / TCP_*_sum derived counters never finished on this pool (15 GPU-minutes lost): SQ counters only.
python3 - <<PY
import csv,glob,collections
for d in ("a",):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv"%d):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0][-30:],r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(agg.items()): print(k, "n=%d avg=%.4g"%(len(v),sum(v)/len(v)))
    import os
    for l in open("$OUT/%s.log"%d):
        if "rror" in l: print(l.strip()[:200])
PY

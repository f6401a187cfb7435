#!/bin/bash
# round 3: packed 16-bit score-only aligner -- parity, then config-3 timing with and without it; then k_ungap L2 counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_al
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "not config4_full and not pipeline and not bench" 2>&1 | tail -5
( timeout 600 python tools/diag/fuzz_parity.py 40 6001 ) > gpurun_out/r03_al/fuzz.log 2>&1; echo "fuzz: $(grep -c ' ok ' gpurun_out/r03_al/fuzz.log) ok"; grep -v " ok " gpurun_out/r03_al/fuzz.log | tail -2
for PK in 1 0; do
  SOHIT_ALIGN_PK=$PK timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-aux --steps 10 --warmup 2 2>/dev/null > gpurun_out/r03_al/c3_pk$PK.json
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03_al/c3_pk$PK.json") if l.startswith('{')][-1])
s=d["stage_ms_per_step"]
print("pk=$PK c3 ms/step", d["ms_per_step"], "rows", d["config"]["rows"], "align_rounds", s.get("phase2.align_rounds"), "trace", s.get("phase2.trace_pass"), "Gcells/s", d["other_kernels"]["k_align_Gcells_per_s"])
PY
done
bash tools/diag/r03_ug_pmc.sh k_ungap r03_ug_pmc --workload c2 2>&1 | tail -20
# HBM traffic of the bucket passes on the weight-6 100k workload (FETCH_SIZE / WRITE_SIZE in their own passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_al/traffic; mkdir -p $OUT; cd $R
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_bkt_pass|k_bkt_group" --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-aux --workload c3w6 > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_bkt_pass|k_bkt_group" --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-aux --workload c3w6 > $OUT/write.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv,glob,collections
for d in ("fetch","write"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv"%d):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0][-40:],r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(agg.items()): print(k, "n=%d avg=%.5g sum=%.5g"%(len(v),sum(v)/len(v),sum(v)))
PY

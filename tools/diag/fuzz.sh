#!/bin/bash
# randomised GPU-vs-oracle differentials (tools/diag/fuzz_parity.py) under the path switches that matter:
#   bash tools/diag/fuzz.sh <cases per mode> <seed> [modes...]      modes: default bucket poison long het hetbucket w64 nobands noug1 nochain wavetb longwave nouq tab hettab eager
N=${1:-30}; SEED=${2:-1}; shift; shift
MODES=${@:-default bucket het hetbucket}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for m in $MODES; do
  case $m in
    default)   E="";;
    bucket)    E="SOHIT_BUCKET_MIN=0";;
    poison)    E="SOHIT_POISON=0xFF SOHIT_BUCKET_MIN=0";;
    long)      E="FUZZ_LONG=1";;
    het)       E="FUZZ_HET=1";;
    hetbucket) E="FUZZ_HET=1 SOHIT_BUCKET_MIN=0 SOHIT_POISON=0xA5";;
    w64)       E="FUZZ_HET=1 SOHIT_BUCKET_MIN=0 SOHIT_UG_W32=0";;
    nobands)   E="FUZZ_HET=1 SOHIT_BUCKET_MIN=0 SOHIT_BANDS=0";;
    noug1)     E="FUZZ_HET=1 SOHIT_BUCKET_MIN=0 SOHIT_UG1=0";;
    nochain)   E="SOHIT_BUCKET_MIN=0 SOHIT_UG1_CHAIN=0";;
    wavetb)    E="FUZZ_HET=1 SOHIT_TRACE_WAVE_ROWS=16 SOHIT_TRACE_WAVE_MAX=100000000 SOHIT_POISON=0x3C";;   # every walk by a wave
    longwave)  E="FUZZ_LONG=1 SOHIT_TRACE_WAVE_ROWS=16 SOHIT_TRACE_WAVE_MAX=100000000";;
    nouq)      E="SOHIT_UNGAPQ=0";;
    tab)       E="SOHIT_BUCKET_MIN=0 SOHIT_COUNT_TAB=2";;   # counts from the range boundaries, checked against the counting pass cell by cell
    eager)     E="FUZZ_HET=1 SOHIT_KSC_LAZY=0";;   # k-mer orders with the batch (default: when the first query reaches its cap)
    hettab)    E="FUZZ_HET=1 SOHIT_BUCKET_MIN=0 SOHIT_COUNT_TAB=2 SOHIT_POISON=0x5A";;
  esac
  env $E python3 tools/diag/fuzz_parity.py $N $SEED > gpurun_out/fuzz_$m.log 2>&1
  echo "$m rc=$? ok=$(grep -c ' ok ' gpurun_out/fuzz_$m.log) fail=$(grep -c FAIL gpurun_out/fuzz_$m.log)"
  grep -B3 FAIL gpurun_out/fuzz_$m.log | tail -8
done

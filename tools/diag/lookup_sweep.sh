#!/bin/bash
# k_lookup tuning / ablation sweep on BASELINE config 2 (GPU box).  Prints avg launch ms per setting.
cd "$(dirname "$0")/../.."
run() { env "$@" python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['ms_per_step'])"; }
run SOHIT_LK_ITERS=8
run SOHIT_LK_ITERS=16
run SOHIT_LK_ITERS=4
run SOHIT_LK_VARIANT=1
run SOHIT_LK_VARIANT=2
run SOHIT_LK_VARIANT=3
run SOHIT_LK_WIDE=1

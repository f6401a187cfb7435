#!/bin/bash
# segmented key sort configuration sweep on BASELINE config 2 (GPU box)
cd "$(dirname "$0")/../.."
run() { env "$@" python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'sort_keys', d['stage_ms_per_step']['group.sort_keys'], 'step', d['ms_per_step'], 'rows', d['config']['rows'])"; }
run SOHIT_SEG_CFG=0
run SOHIT_SEG_CFG=1
run SOHIT_SEG_CFG=2
run SOHIT_SEG_CFG=4
run SOHIT_SEG_CFG=5
run SOHIT_SEG_CFG=6
run SOHIT_SEG_CFG=7

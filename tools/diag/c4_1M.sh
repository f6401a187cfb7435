#!/bin/bash
# BASELINE config-4 workload (1 M proteins x 300 aa, seed 111111) on ONE GPU with the stage clocks on, and a sampled parity check
# (32 queries against the whole reference through the oracle).  usage: bash tools/diag/c4_1M.sh <tag>
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
REPS=1 timeout 1500 python tools/diag/run_config.py 1000000 111111 500000 500032 > gpurun_out/$TAG/c4_1M.txt 2>&1
tail -6 gpurun_out/$TAG/c4_1M.txt | cut -c1-1200

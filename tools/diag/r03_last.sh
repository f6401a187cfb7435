#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_last
REPS=1 timeout 1500 python tools/diag/run_config.py 1000000 111111 500000 500032 > gpurun_out/r03_last/c4_1M.txt 2>&1; tail -6 gpurun_out/r03_last/c4_1M.txt | cut -c1-1200

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_b6
tools/ubench/xdrop | tee gpurun_out/r03_b6/xdrop.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "splits" 2>&1 | tail -5
bash tools/diag/profile_bench.sh r03_b_c3only statsonly --no-aux 2>&1 | tail -2

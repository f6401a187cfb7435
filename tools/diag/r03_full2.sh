#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_full2
( time timeout 2400 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -6
python tools/diag/r03_c5_stages.py 100000 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r03_full2/c5.txt

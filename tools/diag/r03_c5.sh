#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_prof
timeout 1200 python -m pytest tests/test_pipeline.py tests/test_find_orth.py tests/test_abi.py -q -x 2>&1 | tail -3
python tools/diag/r03_c5_stages.py 100000 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r03_prof/c5.txt; cat gpurun_out/r03_prof/c5.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/diag/r03_profiles.sh 2>&1 | tail -3
python tools/diag/r03_c5_stages.py 100000 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r03_prof/c5.txt; cat gpurun_out/r03_prof/c5.txt

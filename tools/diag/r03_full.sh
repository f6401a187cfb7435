#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_full
./tools/ubench/valu > gpurun_out/r03_full/valu.txt 2>&1; grep "8 chains" gpurun_out/r03_full/valu.txt | awk '{print $1, $(NF-6), $(NF-5)}' | head -40
( time timeout 2400 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2

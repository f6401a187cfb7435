#!/bin/bash
# heterogeneous-length parity cases, then the two heterogeneous bench workloads
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_hetlen.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/het_check.log
cat gpurun_out/het_check.log

#!/bin/bash
# heterogeneous-length parity cases, then the two heterogeneous bench workloads (with / without the side-stream k-mer order)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_hetlen.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/het_check.log
bash tools/diag/env_sweep.sh c3het "SOHIT_KSC_ASYNC=0" "SOHIT_KSC_ASYNC=1" >> gpurun_out/het_check.log 2>&1
cat gpurun_out/het_check.log

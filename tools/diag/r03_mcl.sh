#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_find_cluster.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_pipeline.py -x -q -m gpu 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "config4_full or chain or plain_command" 2>&1 | tail -8

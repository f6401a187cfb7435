#!/bin/bash
# round 3 evidence run: driver command, rocprofv3 kernel tables (default workload, config-3 only, config 2), HBM traffic of the roofline workload
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_prof
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_prof/driver_cmd.json 2> gpurun_out/r03_prof/driver_cmd.err
tail -c 600 gpurun_out/r03_prof/driver_cmd.json | head -c 300; echo
bash tools/diag/profile_bench.sh r03_g_c3 statsonly 2>&1 | tail -2
bash tools/diag/profile_bench.sh r03_g_c3only statsonly --no-aux 2>&1 | tail -2
bash tools/diag/profile_bench.sh r03_g_c2 statsonly --workload c2 2>&1 | tail -2
bash tools/diag/profile_bench.sh r03_g_c3w6 full --workload c3w6 2>&1 | tail -2

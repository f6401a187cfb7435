#!/bin/bash
# end-of-round evidence: kernel tables (rocprofv3 --kernel-trace --stats), PMC traffic passes, SQ counters of k_ungap, the driver's
# bench command, config-5 stage times.  usage: bash tools/diag/evidence.sh <tag>      (outputs under gpurun_out/)
T=${1:-r06_a}
cd $GRAFT_REPO_ROOT
bash tools/diag/profile_bench.sh ${T}_c3 full --no-het > gpurun_out/${T}_c3.log 2>&1
bash tools/diag/profile_bench.sh ${T}_c3only statsonly --no-aux > gpurun_out/${T}_c3only.log 2>&1
bash tools/diag/profile_bench.sh ${T}_c3w6 full --workload c3w6 --no-aux > gpurun_out/${T}_c3w6.log 2>&1
bash tools/diag/profile_bench.sh ${T}_c2 statsonly --workload c2 --no-aux > gpurun_out/${T}_c2.log 2>&1
bash tools/diag/profile_bench.sh ${T}_c3het statsonly --workload c3het --no-aux > gpurun_out/${T}_c3het.log 2>&1
bash tools/diag/profile_bench.sh ${T}_c3w6het full --workload c3w6het --no-aux > gpurun_out/${T}_c3w6het.log 2>&1
bash tools/diag/pmc_kernel.sh k_ungap ${T}_sq_ungap --workload c3w6 > gpurun_out/${T}_sq_ungap.txt 2>&1
bash tools/diag/pmc_kernel.sh "k_align|k_traceback" ${T}_sq_align --workload c3 > gpurun_out/${T}_sq_align.txt 2>&1
python3 tools/diag/c5_stages.py > gpurun_out/${T}_c5_stages.txt 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_driver_cmd.json 2> gpurun_out/${T}_driver_cmd.err
python3 tools/diag/bench_summary.py gpurun_out/${T}_driver_cmd.json
tail -12 gpurun_out/${T}_c5_stages.txt

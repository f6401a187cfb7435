#!/bin/bash
# env sweep over one workload: tools/diag/env_sweep.sh <workload> "VAR=a VAR2=b" "VAR=c" ...
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
wl=$1; shift
for cfg in "$@"; do
  echo "== $wl $cfg"
  env $cfg timeout 600 python bench.py --workload $wl --steps 3 --warmup 1 --no-aux --no-cpu-baseline > gpurun_out/sweep_b.json 2>gpurun_out/sweep_b.err
  python tools/diag/bench_summary.py gpurun_out/sweep_b.json | head -2
done > gpurun_out/env_sweep.log 2>&1
cat gpurun_out/env_sweep.log

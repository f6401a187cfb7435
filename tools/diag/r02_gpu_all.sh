# round-2 GPU pass: parity suite, bench (N=1 and the 2-rank functional flow), kernel-trace profile of the bench
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02/pytest.log 2>&1
tail -5 gpurun_out/r02/pytest.log
( time timeout 900 python bench.py ) > gpurun_out/r02/bench_n1.log 2>&1
tail -c 6000 gpurun_out/r02/bench_n1.log
( time SOHIT_BENCH_BACKEND=gloo SOHIT_BENCH_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 ) > gpurun_out/r02/bench_n2_gloo.log 2>&1
tail -c 3000 gpurun_out/r02/bench_n2_gloo.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02/prof -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02/prof.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/r02/prof | head -20

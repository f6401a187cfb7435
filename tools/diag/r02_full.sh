# full round-2 GPU pass: every -m gpu test, randomised differentials, the driver's bench command, the 2-rank flow
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/full/pytest.log 2>&1
tail -6 gpurun_out/full/pytest.log
( timeout 900 python tools/diag/fuzz_parity.py 80 777 ) > gpurun_out/full/fuzz.log 2>&1; grep -c " ok " gpurun_out/full/fuzz.log; grep -v " ok " gpurun_out/full/fuzz.log | tail -3
( SOHIT_BUCKET_MIN=0 timeout 900 python tools/diag/fuzz_parity.py 60 778 ) > gpurun_out/full/fuzz_bkt.log 2>&1; grep -c " ok " gpurun_out/full/fuzz_bkt.log; grep -v " ok " gpurun_out/full/fuzz_bkt.log | tail -3
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/full/bench_n1.log 2>&1
tail -c 5000 gpurun_out/full/bench_n1.log
( SOHIT_BENCH_BACKEND=gloo SOHIT_BENCH_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 ) > gpurun_out/full/bench_n2_gloo.log 2>&1
tail -c 1500 gpurun_out/full/bench_n2_gloo.log

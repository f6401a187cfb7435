# extra hardening runs: long sequences, poisoned allocations, forced bucket path
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
( FUZZ_LONG=1 timeout 900 python tools/diag/fuzz_parity.py 40 4242 ) > gpurun_out/final/fuzz_long.log 2>&1; echo "long: $(grep -c ' ok ' gpurun_out/final/fuzz_long.log) ok"; grep -v " ok " gpurun_out/final/fuzz_long.log | tail -2
( SOHIT_POISON=0xFF SOHIT_BUCKET_MIN=0 timeout 900 python tools/diag/fuzz_parity.py 60 4243 ) > gpurun_out/final/fuzz_poison_bkt.log 2>&1; echo "poison+bucket: $(grep -c ' ok ' gpurun_out/final/fuzz_poison_bkt.log) ok"; grep -v " ok " gpurun_out/final/fuzz_poison_bkt.log | tail -2
( SOHIT_POISON=0x5A SOHIT_BUCKET_MIN=0 SOHIT_BUCKET_AVG=100000 timeout 900 python tools/diag/fuzz_parity.py 40 4244 ) > gpurun_out/final/fuzz_poison_wide.log 2>&1; echo "poison+wide buckets: $(grep -c ' ok ' gpurun_out/final/fuzz_poison_wide.log) ok"; grep -v " ok " gpurun_out/final/fuzz_poison_wide.log | tail -2
( SOHIT_BUCKET_MIN=0 SOHIT_BUCKET_AVG=48 timeout 900 python tools/diag/fuzz_parity.py 40 4245 ) > gpurun_out/final/fuzz_narrow.log 2>&1; echo "narrow buckets: $(grep -c ' ok ' gpurun_out/final/fuzz_narrow.log) ok"; grep -v " ok " gpurun_out/final/fuzz_narrow.log | tail -2

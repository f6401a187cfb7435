cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bkt
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or synth or uniform or long or bucketed or ragged or tandem or ties" ) > gpurun_out/bkt/pytest_ug.log 2>&1
tail -2 gpurun_out/bkt/pytest_ug.log
timeout 600 python bench.py --workload c2 --no-cpu-baseline --no-aux --steps 10 --warmup 2 > gpurun_out/bkt/ug_c2.json 2> gpurun_out/bkt/ug_c2.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bkt/ug_c2.json") if l.startswith("{")][-1])
print("c2 ms_per_step", d["ms_per_step"], d["stage_ms_per_step"]["group.ungap"])
PY
bash tools/diag/pmc_kernel.sh "k_ungap" pmc_ug --workload c2 2>&1 | grep "CONFLICT\|IDX_ACTIVE\|INSTS_VALU'\|WAIT_INST_LDS"

#!/bin/bash
# how much of a step the GPU spends with NO kernel running (host round trips, launch gaps): union of the kernel intervals of a
# rocprofv3 --kernel-trace run of bench.py, over the span of the timed steps.   bash tools/diag/gpu_idle.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/gpu_idle; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-aux "$@" > $OUT.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*kernel_trace.csv")[0]
ev = []
for r in csv.DictReader(open(f)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
ev.sort()
# steps: the index build starts every step: k_index_windows<false> launches (2 chunks per step -> every second one)
starts = [s for s, e, n in ev if "k_index_windows<false>" in n][::2]
print("steps seen", len(starts))
for a, b in list(zip(starts, starts[1:]))[2:6]:
    busy, cur_s, cur_e, nk = 0, None, None, 0
    gaps = []
    for s, e, n in ev:
        if e <= a or s >= b: continue
        nk += 1
        s, e = max(s, a), min(e, b)
        if cur_e is None: cur_s, cur_e = s, e
        elif s <= cur_e: cur_e = max(cur_e, e)
        else:
            gaps.append((s - cur_e, n)); busy += cur_e - cur_s; cur_s, cur_e = s, e
    busy += cur_e - cur_s
    big = sorted(gaps, reverse=True)[:6]
    print("step %.2f ms: busy %.2f ms, idle %.2f ms in %d gaps (%d kernels); gaps > 20 us: %d = %.2f ms; largest before: %s" % (
        (b - a) / 1e6, busy / 1e6, (b - a - busy) / 1e6, len(gaps), nk, sum(1 for g, _ in gaps if g > 20000), sum(g for g, _ in gaps if g > 20000) / 1e6,
        ", ".join("%s %.0fus" % (n, g / 1e3) for g, n in big)))
PY
rm -rf $OUT

# GPU idle gaps between kernels of a bench run: bash tools/diag/r02_gaps.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/gaps; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aux "$@" > $OUT/log 2>&1
python3 - <<PY
import csv, glob
ev = []
for f in glob.glob("$OUT/t/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
for f in glob.glob("$OUT/t/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "")))
ev.sort()
# the timed steps: take the last 60 % of the trace (warm-up and set-up come first)
t0, t1 = ev[0][0], ev[-1][1]
cut = t0 + (t1 - t0) * 0.4
ev = [e for e in ev if e[0] >= cut]
busy = 0; end = ev[0][0]; gaps = []
for s, e, n in ev:
    if s > end:
        gaps.append((s - end, n)); end_prev = end
    if e > end:
        busy += e - max(s, end); end = e
span = ev[-1][1] - ev[0][0]
print("span %.2f ms busy %.2f ms idle %.2f ms (%.1f %%) events %d" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, 100 * (span - busy) / span, len(ev)))
import collections
big = sorted(gaps, reverse=True)[:12]
print("largest gaps (us, next event):", [(round(g / 1e3, 1), n) for g, n in big])
hist = collections.Counter(min(int(g / 1e4), 20) for g, _ in gaps)
print("gap histogram (10 us bins):", sorted(hist.items()))
by = collections.defaultdict(float)
for g, n in gaps: by[n] += g
print("idle before:", [(n, round(v / 1e6, 2)) for n, v in sorted(by.items(), key=lambda x: -x[1])[:14]])
PY
find $OUT -name "*.csv" -delete

#!/usr/bin/env python3
"""print the top kernels of a rocprofv3 --kernel-trace --stats run: python tools/diag/top_kernels.py <dir with *_kernel_stats.csv> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total %.1f ms in %d kernels" % (tot / 1e6, len(rows)))
for r in rows[:n]:
    print("%-100s %6s calls %10.2f ms total %10.1f us avg" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))

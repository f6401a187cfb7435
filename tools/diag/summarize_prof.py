#!/usr/bin/env python3
"""Distil gpurun_out/prof_<tag>/ (written by tools/diag/profile_c2.sh) into profiles/:
  profiles/<tag>_kernel_stats_c2.csv   rocprofv3 --kernel-trace --stats summary (copied as is)
  profiles/traffic_r01.json            per-kernel FETCH_SIZE / WRITE_SIZE (KiB per dispatch) + the k_lookup HBM bytes
                                       bench.py reports as roofline.traffic
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-byte fabric read requests at 64 bytes,
so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact.
usage: python tools/diag/summarize_prof.py r01_d
"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_d"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)


def one(pattern):
    f = glob.glob(os.path.join(src, pattern))
    if not f:
        raise SystemExit("missing " + pattern)
    return f[0]


shutil.copy(one("stats/*/*_kernel_stats.csv"), os.path.join(ROOT, "profiles", tag + "_kernel_stats_c2.csv"))


def pmc(path, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"]
            k = k.split("(")[0].replace("void ", "")
            if "rocprim" in k:
                k = "rocprim::" + k.split("::")[-1][:40]
            agg[k].append(float(r["Counter_Value"]))
    return {k: round(sum(v) / len(v), 1) for k, v in agg.items() if sum(v) / len(v) > 1024}


fetch = pmc(one("fetch/*/*_counter_collection.csv"), "FETCH_SIZE")
write = pmc(one("write/*/*_counter_collection.csv"), "WRITE_SIZE")
lk = [k for k in fetch if k.startswith("k_lookup<")][0]
bench = json.loads([l for l in open(os.path.join(src, "stats.log")) if l.startswith("{")][-1])
H = bench["other_kernels"]["seed_hits_per_step"]
rd, wr = int(2 * fetch[lk] * 1024), int(write[lk] * 1024)
out = {
    "_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 2 --warmup 1 "
            "--no-cpu-baseline` (BASELINE config 2, one MI355X; tools/diag/profile_c2.sh " + tag + "). Counters are KiB per dispatch. "
            "gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-byte fabric read requests at 64 bytes, so read "
            "bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact (k_lookup writes exactly 8 B x H keys). The kernel reads 4-byte "
            "compact index addends (4 B x H) of a 12 MB table: part of those reads hit the XCD L2 and never reach the fabric.",
    "H_seed_hits_per_launch": H,
    "k_lookup_kernel": lk,
    "k_lookup_fetch_size_kib": fetch[lk],
    "k_lookup_write_size_kib": write[lk],
    "k_lookup_hbm_read_bytes_per_launch": rd,
    "k_lookup_hbm_write_bytes_per_launch": wr,
    "k_lookup_hbm_bytes_per_launch": rd + wr,
    "k_lookup_algorithmic_read_bytes_per_launch": 8 * H,
    "kernels_kib_per_dispatch": {k: {"FETCH_SIZE": fetch.get(k), "WRITE_SIZE": write.get(k)} for k in sorted(set(fetch) | set(write))},
}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_r01.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.startswith("k_lookup") or k.startswith("H_")}, indent=1))

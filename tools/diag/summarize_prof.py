#!/usr/bin/env python3
"""Distil gpurun_out/prof_<tag>/ (written by tools/diag/profile_bench.sh) into profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (copied as is)
  profiles/<tag>_bench.json         the bench line of that run
  profiles/traffic_<round>.json     (only when the PMC passes exist) per-kernel FETCH_SIZE / WRITE_SIZE per dispatch and the
                                    seed-lookup kernel's HBM bytes that bench.py quotes as roofline.traffic
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-byte fabric read requests at 64 bytes, so
read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact.
usage: python tools/diag/summarize_prof.py <tag> "<workload description>" [name of the traffic file under profiles/]
"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else ""
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)


def one(pattern):
    f = glob.glob(os.path.join(src, pattern))
    return f[0] if f else None


shutil.copy(one("stats/*/*_kernel_stats.csv"), os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
bench = json.loads([l for l in open(os.path.join(src, "stats.log")) if l.startswith("{")][-1])
json.dump(bench, open(os.path.join(ROOT, "profiles", tag + "_bench.json"), "w"), indent=1)


def pmc(path, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "rocprim" in k:
                k = "rocprim::" + k.split("::")[-1][:40]
            agg[k].append(float(r["Counter_Value"]))
    return {k: round(sum(v) / len(v), 1) for k, v in agg.items() if sum(v) / len(v) > 1024}


if one("fetch/*/*_counter_collection.csv") and one("write/*/*_counter_collection.csv"):
    fetch = pmc(one("fetch/*/*_counter_collection.csv"), "FETCH_SIZE")
    write = pmc(one("write/*/*_counter_collection.csv"), "WRITE_SIZE")
    # seed hits of ONE launch of the lookup kernel (a step holds several launches when the batch is cut into hit-budgeted passes)
    H = bench["roofline"]["algorithmic_bytes_per_launch"] // 8 if bench.get("roofline", {}).get("algorithmic_bytes_per_launch") else bench["other_kernels"]["seed_hits_per_step"]
    out = {
        "_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 1 --warmup 1 "
                "--no-cpu-baseline` + the workload flags (tools/diag/profile_bench.sh " + tag + "). Counters are KiB per dispatch. gfx950 "
                "correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-byte fabric read requests at 64 bytes, so read bytes = "
                "2 x FETCH_SIZE x 1024; WRITE_SIZE is exact.",
        "workload": workload,
        "H_seed_hits_per_launch": H,
        "algorithmic_read_bytes_per_launch": 8 * H,
        "kernels_kib_per_dispatch": {k: {"FETCH_SIZE": fetch.get(k), "WRITE_SIZE": write.get(k)} for k in sorted(set(fetch) | set(write))},
    }
    def pick(prefix):   # kernel names carry their template arguments: k_bkt_pass<true, true, 5> ...
        c = [k for k in sorted(set(fetch) | set(write)) if k.startswith(prefix)]
        return c[0] if c else None
    for name, key in ((pick("k_bkt_pass<true"), "scatter"), (pick("k_bkt_count_tab") or pick("k_bkt_pass<false"), "count"), (pick("k_bkt_group"), "group"), (pick("k_lookup<16"), "k_lookup")):
        if name and (name in fetch or name in write):
            rd, wr = int(2 * fetch.get(name, 0) * 1024), int(write.get(name, 0) * 1024)
            out[key] = {"kernel": name, "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                        "bytes_per_hit": round((rd + wr) / H, 3), "over_algorithmic": round((rd + wr) / (8 * H), 3)}
    if "scatter" in out:
        out["k_lookup_hbm_bytes_per_launch"] = out["scatter"]["hbm_bytes"]   # the kernel bench.py's roofline object names
        out["binning_hbm_bytes_per_launch"] = sum(out[k]["hbm_bytes"] for k in ("count", "scatter", "group") if k in out)
        out["binning_over_algorithmic"] = round(out["binning_hbm_bytes_per_launch"] / (8 * H), 3)
    json.dump(out, open(os.path.join(ROOT, "profiles", sys.argv[3] if len(sys.argv) > 3 else "traffic_%s.json" % tag.split("_")[0]), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k in ("scatter", "count", "group", "binning_over_algorithmic")}, indent=1))

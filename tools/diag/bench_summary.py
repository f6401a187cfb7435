#!/usr/bin/env python3
"""one-screen summary of a bench.py JSON line: python tools/diag/bench_summary.py <file>"""
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("headline %.3f ms  %.1f M aa/s  first %.1f  nocache %.1f" % (j["ms_per_step"], j["value"], j.get("ms_first_step") or 0, j.get("ms_per_step_hit_cache_off") or 0))
print("  stages", j["stage_ms_per_step"])
r = j["roofline"]
print("  roofline frac %.4f (%.4f ms)  count %s" % (r["frac"], r["avg_launch_ms"], (j.get("roofline_count_pass") or {}).get("frac")))
if "strong_scaling_aux" in j:
    a = j["strong_scaling_aux"]
    print("w6 uniform %.1f ms  %.2f M aa/s" % (a["ms_per_step"], a["value"]))
    print("  stages", a["stage_ms_per_step"])
for k, v in (j.get("length_heterogeneous_aux") or {}).items():
    if "ms_per_step" in v:
        print("het %s %.1f ms  %.2f M aa/s  bucketed %.3f packed %.3f" % (k, v["ms_per_step"], v["value"], v["seed_hits_through_bucketed_passes"], v["score_only_cells_through_packed_aligner"]))
        print("  stages", v["stage_ms_per_step"])
if "cpu_baseline" in j:
    print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])

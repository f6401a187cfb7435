"""Container-only: golden for the end-to-end pipeline of BASELINE config 5
(find_hit -> find_orth -> find_cluster -a mcl -I 1.5) at a size the fixtures afford.

    python tools/refharness/make_pipeline_golden.py c2|c3 [--force]

The search rows come from oracle/sohit_cpu (P processes over query blocks, concatenated: what the reference launcher
does), the two downstream stages are the REAL reference scripts (bin/find_orth.py; bin/find_cluster.py with the numba /
cffi import stand-ins of tools/refharness/fcshim).  The input proteome is regenerated from swiftortho_amd.synthprot on
the test side, so only digests and the final groups are stored:
  tests/golden/pipe_<cfg>.json     flags, md5 of the .sc / .orth, row counts, sha256 of the canonical groups
  tests/golden/pipe_<cfg>.groups   the ortholog groups (c2 only; for c3 the digest + a 200-line head)
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
CFG = {"c2": (10000, "111111"), "c3": (100000, "11111011111")}


def canonical(groups_text):
    """scripts/mcl_cmp.py compares clusterings as sets of gene sets"""
    rows = sorted("\t".join(sorted(l.split("\t"))) for l in groups_text.split("\n") if l)
    return "\n".join(rows) + "\n"


def main():
    import make_cluster_goldens as mc
    import make_orth_goldens as mo
    from oracle import oracle
    from swiftortho_amd import synthprot
    cfg = sys.argv[1]
    n, ssd = CFG[cfg]
    oracle.build()
    fa = synthprot.synthprot(n, 300)
    flags = ["-e", "1e-5", "-v", "500", "-j", "1", "-F", "T", "-s", ssd, "-r", oracle.AA9, "-M", "120000000", "-c", "50000"]
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "x.fsa")
        open(p, "wb").write(fa)
        P = os.cpu_count() or 1
        block = (n + P - 1) // P
        t0 = time.time()
        procs = []
        for k in range(P):
            lo, hi = k * block, min(n, (k + 1) * block)
            procs.append(subprocess.Popen([oracle.EXE, "-p", "blastp", "-i", p, "-d", p, "-o", os.path.join(d, "%03d.sc" % k), "-l", str(lo), "-u", str(hi)] + flags,
                                          stderr=subprocess.DEVNULL))
        for q in procs:
            assert q.wait() == 0
        sc = os.path.join(d, "all.sc")
        with open(sc, "wb") as o:
            for k in range(P):
                o.write(open(os.path.join(d, "%03d.sc" % k), "rb").read())
        print("search %.0fs rows %d" % (time.time() - t0, open(sc, "rb").read().count(b"\n")))
        t0 = time.time()
        orth = mo.run_ref_find_orth(sc, [])
        print("find_orth %.0fs lines %d" % (time.time() - t0, orth.count(b"\n")))
        op = os.path.join(d, "all.orth")
        open(op, "wb").write(orth)
        t0 = time.time()
        groups = mc.run_ref_find_cluster(op, ["-a", "mcl", "-I", "1.5"]).decode()
        print("find_cluster %.0fs groups %d" % (time.time() - t0, groups.count("\n")))
        meta = {"proteins": n, "find_hit_flags": flags, "find_orth_flags": [], "find_cluster_flags": ["-a", "mcl", "-I", "1.5"],
                "fasta_md5": hashlib.md5(fa).hexdigest(), "sc_md5": hashlib.md5(open(sc, "rb").read()).hexdigest(),
                "sc_rows": open(sc, "rb").read().count(b"\n"), "orth_md5": hashlib.md5(orth).hexdigest(), "orth_lines": orth.count(b"\n"),
                "groups": groups.count("\n"), "genes_in_groups": len(groups.split()),
                "groups_text_sha256": hashlib.sha256(groups.encode()).hexdigest(),
                "groups_canonical_sha256": hashlib.sha256(canonical(groups).encode()).hexdigest()}
        json.dump(meta, open(os.path.join(GOLD, "pipe_%s.json" % cfg), "w"), indent=1)
        body = groups if cfg == "c2" else "".join(l + "\n" for l in groups.split("\n")[:200])
        open(os.path.join(GOLD, "pipe_%s.groups" % cfg), "w").write(body)
        print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()

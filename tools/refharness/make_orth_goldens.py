"""Container-only: goldens for the find_orth counterpart.

    python tools/refharness/make_orth_goldens.py [--force]

Runs the REAL reference script /root/reference/bin/find_orth.py (plain Python 3 + GNU sort) on .sc files
and stores its stdout.  Inputs are data: multi-taxon synthetic proteomes searched by oracle/sohit_cpu (any
16-column .sc is a valid input of find_orth; what is pinned here is find_orth, not the search).
Fixtures: tests/golden/orth_<name>.sc (input), orth_<name>.<variant>.orth (expected stdout) and
orth_<name>.json (flags per variant).  No reference source text is stored.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REFERENCE = os.environ.get("SWIFTORTHO_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
FORCE = "--force" in sys.argv
AA = "ARNDCQEGHILKMFPSTWYV"

VARIANTS = {"default": [], "cov70_idy30": ["-c", "0.7", "-y", "30"], "bsr": ["-n", "bsr"], "bal": ["-n", "bal", "-c", "0.3"]}


def taxa_fasta(n_fam, n_taxa, L, seed, p_copy=0.8, p_dup=0.35, sep="|"):
    """n_fam families x n_taxa taxa: every taxon gets a diverged copy with p_copy, and with p_dup one or two
    in-paralogs (recent duplicates: low divergence from that copy), so that IP, OT and CO relations all occur."""
    rng = np.random.default_rng(seed)

    def mutate(s, d):
        s = list(s)
        for i in range(len(s)):
            if rng.random() < d:
                s[i] = AA[int(rng.integers(0, 20))]
        if rng.random() < 0.5:
            p = int(rng.integers(5, len(s) - 5))
            if rng.random() < 0.5:
                del s[p:p + int(rng.integers(1, 4))]
            else:
                s[p:p] = [AA[int(rng.integers(0, 20))]] * int(rng.integers(1, 4))
        return "".join(s)

    per_taxon = [[] for _ in range(n_taxa)]
    for f in range(n_fam):
        anc = "".join(AA[i] for i in rng.integers(0, 20, L))
        for t in range(n_taxa):
            if rng.random() < p_copy:
                c = mutate(anc, rng.uniform(0.05, 0.45))
                per_taxon[t].append(c)
                if rng.random() < p_dup:
                    for _ in range(int(rng.integers(1, 3))):
                        per_taxon[t].append(mutate(c, rng.uniform(0.02, 0.15)))
    out, k = [], 0
    for t in range(n_taxa):
        for s in per_taxon[t]:
            out.append(">tx%02d%sg%05d some description\n%s\n" % (t, sep, k, s))
            k += 1
    return "".join(out).encode()


def run_ref_find_orth(sc_path, flags):
    """stdout of the reference script; it litters <input>_tmp/ and ./tmp/, so run inside a scratch dir"""
    with tempfile.TemporaryDirectory(prefix="orth_") as d:
        local = os.path.join(d, "in.sc")
        open(local, "wb").write(open(sc_path, "rb").read())
        r = subprocess.run([sys.executable, os.path.join(REFERENCE, "bin", "find_orth.py"), "-i", local] + flags, cwd=d,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, LC_ALL="C"))
        if r.returncode != 0:
            raise RuntimeError(r.stderr.decode()[-2000:])
        return r.stdout


def make(name, fasta, search_flags, sep="|"):
    from oracle import oracle
    oracle.build()
    sc = os.path.join(GOLD, "orth_%s.sc" % name)
    if os.path.isfile(sc) and not FORCE:
        print(name, "exists, skipped")
        return
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "x.fsa")
        open(fa, "wb").write(fasta)
        subprocess.run([oracle.EXE, "-p", "blastp", "-i", fa, "-d", fa, "-o", sc] + search_flags, check=True, stderr=subprocess.DEVNULL)
    meta = {"variants": {}, "sep": sep}
    for v, fl in VARIANTS.items():
        fl = fl + (["-s", sep] if sep != "|" else [])
        out = run_ref_find_orth(sc, fl)
        open(os.path.join(GOLD, "orth_%s.%s.orth" % (name, v)), "wb").write(out)
        meta["variants"][v] = fl
        kinds = {k: sum(1 for l in out.split(b"\n") if l.startswith(k)) for k in (b"IP", b"OT", b"CO")}
        print(name, v, "rows", open(sc, "rb").read().count(b"\n"), {k.decode(): n for k, n in kinds.items()})
    json.dump(meta, open(os.path.join(GOLD, "orth_%s.json" % name), "w"), indent=1)


def main():
    base = ["-e", "1e-5", "-v", "500", "-j", "1", "-F", "T", "-s", "111111", "-M", "1000003", "-c", "50000"]
    make("taxa5", taxa_fasta(60, 5, 110, 11), base)
    make("taxa3_dense", taxa_fasta(40, 3, 90, 12, p_copy=0.95, p_dup=0.6), base)
    make("taxa4_colon", taxa_fasta(30, 4, 100, 13, sep=":"), base, sep=":")
    # an existing search golden (two taxa) as well
    if FORCE or not os.path.isfile(os.path.join(GOLD, "orth_toy_default.json")):
        sc = os.path.join(GOLD, "toy_default.sc")
        meta = {"variants": {}, "sep": "|", "input": "toy_default.sc"}
        for v, fl in VARIANTS.items():
            open(os.path.join(GOLD, "orth_toy_default.%s.orth" % v), "wb").write(run_ref_find_orth(sc, fl))
            meta["variants"][v] = fl
        json.dump(meta, open(os.path.join(GOLD, "orth_toy_default.json"), "w"), indent=1)
        print("toy_default done")


if __name__ == "__main__":
    main()

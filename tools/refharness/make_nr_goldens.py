"""Container-only: golden for the duplicate-expansion helper.

    python tools/refharness/make_nr_goldens.py

A proteome with exact duplicates is collapsed by the REAL /root/reference/scripts/nr_flt.py -- run with
tools/refharness/biostub on PYTHONPATH, a stand-in for the one Biopython call it makes (Bio.SeqIO.parse of PLAIN FASTA, on
which every FASTA parser agrees: the golden pins nr_flt.py's own grouping / ordering / joining, not Biopython) --,
searched by oracle/sohit_cpu, and the REAL /root/reference/scripts/nr2full.py (stdlib only) expands the hits.
Fixtures: tests/golden/nr_dups.fsa (input proteome), nr_dups.nr.fsa (expected stdout of nr_flt), nr_dups.nr.sc
(collapsed search = input of nr2full), nr_dups.full.sc (expected stdout of nr2full).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REFERENCE = os.environ.get("SWIFTORTHO_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    import make_orth_goldens as mo
    from oracle import oracle
    from swiftortho_amd import nr
    oracle.build()
    rng = np.random.default_rng(5)
    recs = [l for l in mo.taxa_fasta(25, 3, 90, 31).decode().split("\n") if l]
    pairs = [(recs[i], recs[i + 1]) for i in range(0, len(recs), 2)]
    extra = []
    for k in range(18):   # exact duplicates under new ids, some of them three times
        hd, sq = pairs[int(rng.integers(0, len(pairs)))]
        extra.append((">tx%02d|dup%03d copy of %s" % (int(rng.integers(0, 3)), k, hd[1:].split()[0]), sq))
    allrecs = pairs + extra
    order = rng.permutation(len(allrecs))
    fasta = "".join("%s\n%s\n" % allrecs[i] for i in order)
    open(os.path.join(GOLD, "nr_dups.fsa"), "w").write(fasta)
    with tempfile.TemporaryDirectory() as d:
        nrfa = os.path.join(GOLD, "nr_dups.nr.fsa")
        env = dict(os.environ, PYTHONPATH=os.path.join(HERE, "biostub"))
        r = subprocess.run([sys.executable, os.path.join(REFERENCE, "scripts", "nr_flt.py"), os.path.join(GOLD, "nr_dups.fsa")], stdout=subprocess.PIPE,
                           check=True, env=env)
        open(nrfa, "wb").write(r.stdout)
        assert r.stdout.decode() == "\n".join(nr.nr_flt(fasta.splitlines(True))) + "\n", "counterpart differs from the reference script"
        sc = os.path.join(GOLD, "nr_dups.nr.sc")
        subprocess.run([oracle.EXE, "-p", "blastp", "-i", nrfa, "-d", nrfa, "-o", sc, "-e", "1e-5", "-v", "500", "-j", "1", "-F", "T", "-s", "111111",
                        "-M", "1000003", "-c", "50000"], check=True, stderr=subprocess.DEVNULL)
        r = subprocess.run([sys.executable, os.path.join(REFERENCE, "scripts", "nr2full.py"), sc], stdout=subprocess.PIPE, check=True)
        open(os.path.join(GOLD, "nr_dups.full.sc"), "wb").write(r.stdout)
        print("collapsed rows", open(sc).read().count("\n"), "expanded rows", r.stdout.count(b"\n"))


if __name__ == "__main__":
    main()

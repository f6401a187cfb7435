"""Length-heterogeneous end-to-end goldens from the REAL reference (container only).

    python tools/refharness/make_het_goldens.py [--force]

Input: synthprot(300, seed=5, lengths="lognormal") -- median ~270 residues, a tail to 5000, subjects of 4562 and 4597 residues
and one of 30 014.  The reference itself is UNDEFINED for a query of 4096+ residues that meets a shorter subject:
kswat_st_long (fsearch.py:1487-1490) cuts the subject tile S1[4096*k : ...] to an empty string, kswat_st's
`sed = sed < 0 and len(S1) or sed` (fsearch.py:1362) then yields -1 and fsearch.py:1396 indexes the empty string
(IndexError under CPython, an out-of-bounds read in the RPython build).  So the reference is run with -l / -u ranges that
cover every query BELOW 4096 residues and skip the longer ones; per-query results do not depend on the range they are
run in (find_hit.py:107-146 relies on the same fact), so the concatenation of the ranges' outputs is the reference's
answer for those queries against the WHOLE reference, giants included as subjects.

Writes only data: het_<seed>.ref.fsa (the FASTA), het_<seed>.sc (the ranges' rows in order), het_<seed>.json (flags, ranges,
the skipped query ordinals).
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refload  # noqa: E402
from swiftortho_amd import synthprot  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
AA9 = "AST,CFILMVY,DN,EQ,G,H,KR,P,W"
FORCE = "--force" in sys.argv


def query_ranges(lengths, limit=4096):
    """maximal [lo, hi) ranges of query ordinals whose sequences are all shorter than `limit`"""
    out, lo = [], 0
    for i, n in enumerate(list(lengths) + [limit]):
        if n >= limit:
            if i > lo:
                out.append([lo, i])
            lo = i + 1
    return out


def run(m, name, fasta, flags):
    if os.path.isfile(os.path.join(GOLD, name + ".sc")) and not FORCE:
        print(name, "exists, skipped")
        return
    ln = np.array([len(x) for x in fasta.split(b"\n")[1::2]])
    ranges = query_ranges(ln)
    skipped = [int(i) for i in np.nonzero(ln >= 4096)[0]]
    tmp = tempfile.mkdtemp(prefix="gold_het_")
    fa = os.path.join(tmp, "ref.fsa")
    open(fa, "wb").write(fasta)
    rows = b""
    for lo, hi in ranges:
        out = os.path.join(tmp, "out_%d.sc" % lo)
        t0 = time.time()
        m.entry_point(["fsearch", "-p", "blastp", "-i", fa, "-d", fa, "-o", out, "-T", tmp, "-l", str(lo), "-u", str(hi)] + flags)
        part = open(out, "rb").read()
        print(name, "range", lo, hi, "rows", part.count(b"\n"), "%.0fs" % (time.time() - t0), flush=True)
        rows += part
    open(os.path.join(GOLD, name + ".ref.fsa"), "wb").write(fasta)
    open(os.path.join(GOLD, name + ".sc"), "wb").write(rows)
    json.dump({"flags": flags, "ranges": ranges, "skipped_queries": skipped, "skipped_lengths": [int(ln[i]) for i in skipped],
               "longest_subject": int(ln.max()), "sequences": int(len(ln)),
               "why_skipped": "fsearch.py:1362/1396/1487-1490: a query tile past the end of a shorter subject indexes an empty string"},
              open(os.path.join(GOLD, name + ".json"), "w"), indent=1)
    print(name, "rows:", rows.count(b"\n"))


def main():
    m = refload.load()
    fa = synthprot.synthprot(300, seed=5, lengths="lognormal")
    base = ["-e", "1e-5", "-v", "500", "-j", "1", "-F", "T", "-r", AA9, "-c", "50000"]
    # -M 1000003: a full-size table costs CPython ~1 GB and tens of seconds per range; collisions are part of the contract anyway
    run(m, "het_w6", fa, base + ["-s", "111111", "-M", "1000003"])
    run(m, "het_w10", fa, base + ["-s", "11111011111", "-M", "1000003"])


if __name__ == "__main__":
    main()

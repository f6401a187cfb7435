"""Exact-duplicate collapse / expand helpers (swiftortho_amd/nr.py) against stdout of the REAL reference scripts
(tools/refharness/make_nr_goldens.py): nr2full.py as it is; nr_flt.py run with a stand-in for its one Biopython call
(Bio.SeqIO.parse of plain FASTA, where every parser agrees).  Hand-written cases and the round trip on top.  CPU only;
the collapse -> GPU search -> expand chain is in test_gpu_parity.py."""
import os
import subprocess
import sys

from conftest import GOLD, ROOT


def test_nr2full_matches_reference_output():
    from swiftortho_amd import nr
    got = nr.nr2full(open(os.path.join(GOLD, "nr_dups.nr.sc")))
    want = open(os.path.join(GOLD, "nr_dups.full.sc")).read().split("\n")[:-1]
    assert len(want) > len(open(os.path.join(GOLD, "nr_dups.nr.sc")).readlines())
    assert got == want
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "nr2full.py"), os.path.join(GOLD, "nr_dups.nr.sc")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == open(os.path.join(GOLD, "nr_dups.full.sc")).read()


def test_nr_flt_matches_reference_output():
    from swiftortho_amd import nr
    want = open(os.path.join(GOLD, "nr_dups.nr.fsa")).read()
    assert want.count(">") < open(os.path.join(GOLD, "nr_dups.fsa")).read().count(">")     # duplicates were merged
    assert "\n".join(nr.nr_flt(open(os.path.join(GOLD, "nr_dups.fsa")))) + "\n" == want
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "nr_flt.py"), os.path.join(GOLD, "nr_dups.fsa")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == want


def test_nr_flt_by_hand():
    from swiftortho_amd import nr
    fa = [">a|1 first protein\n", "MKV\n", "LLA\n", ">b|2\n", "GGG\n", ">c|3 same as a\n", "MKVLLA\n", ">d|4\n", "GGG\n", ">e|5\n", "MKVLLa\n"]
    assert nr.nr_flt(fa) == [">a|1;;;c|3", "MKVLLA", ">b|2;;;d|4", "GGG", ">e|5", "MKVLLa"]   # first-appearance order, ids only, case-sensitive
    assert nr.nr_flt([]) == []
    # Biopython's documented record rules (not pinned by a run of Biopython): text before the first '>' is ignored, blanks and CR
    # inside residue lines are dropped, trailing blanks of the title too, a bare '>' has the empty id
    odd = ["junk\n", ">x|1  two  words \n", "MK V\r\n", "LL\n", ">\n", "MKVLL\n"]
    assert nr.nr_flt(odd) == [">x|1;;;", "MKVLL"]


def test_collapse_search_expand_round_trip():
    """every expanded row names individual ids of the original proteome, and every duplicate of a query gets the same rows"""
    from swiftortho_amd import nr
    fasta = open(os.path.join(GOLD, "nr_dups.fsa")).read().splitlines(True)
    ids = {h.split()[0] for h, _ in nr.fasta_records(fasta)}
    seq_of = {h.split()[0]: s for h, s in nr.fasta_records(fasta)}
    rows = [r.split("\t") for r in nr.nr2full(open(os.path.join(GOLD, "nr_dups.nr.sc")))]
    assert all(r[0] in ids and r[1] in ids and r[-2] == r[0] and r[-1] == r[1] and len(r) == 16 for r in rows)
    by_q = {}
    for r in rows:
        by_q.setdefault(r[0], set()).add(tuple(r[1:14]))
    dup_groups = {}
    for i, s in seq_of.items():
        dup_groups.setdefault(s, []).append(i)
    checked = 0
    for members in dup_groups.values():
        hit = [m for m in members if m in by_q]
        for m in hit[1:]:
            assert by_q[m] == by_q[hit[0]]
            checked += 1
    assert checked >= 10

"""The oracle (oracle/sohit_cpu.cpp) against fixtures produced by the REAL reference source
(tools/refharness/make_goldens.py, run in the build container).  CPU only."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLD, golden_names, het_golden_names, launcher_golden_names


@pytest.fixture(scope="module")
def kat():
    return json.load(open(os.path.join(GOLD, "kat.json")))


def test_b62_table(oracle, kat):
    L = kat["b62_letters"]
    for a, row in zip(L, kat["b62_23x23"]):
        for b, v in zip(L, row):
            assert oracle.b62(a, b) == v
            assert oracle.b62(a.lower(), b) == v and oracle.b62(a, b.lower()) == v
    assert oracle.b62("U", "U") == kat["b62_default"]
    for a, b, v in kat["b62_probe"]:
        assert oracle.b62(a, b) == v
    assert int(oracle.b62_matrix().sum()) == kat["b62_sum"]


def test_nr_tables(oracle, kat):
    for g, tbl in kat["nr_tbl"].items():
        assert oracle.nr_tbl(g) == tbl


def test_spseeds(oracle, kat):
    for c in kat["spseeds"]:
        got = oracle.spseeds(c["seq"], c["ssd"], c["nr"], c["mod"], c["step"])
        assert got == [tuple(x) for x in c["out"]], c


def test_seg(oracle, kat):
    for c in kat["seg"]:
        assert oracle.seg(c["in"]).decode() == c["out"], c["in"]


def test_qsort(oracle, kat):
    for c in kat["qsort"]:
        assert oracle.qsort_perm(c["keys"]) == c["perm"]


def test_ungap(oracle, kat):
    for c in kat["ungap"]:
        assert list(oracle.ungap(c["q"], c["s"], c["Qst"], c["Sst"], c["qlo"], c["slo"])) == c["out"], c
    for c in kat["ungap_chain"]:
        assert list(oracle.ungap_chain(c["q"], c["s"], c["locs"])) == c["out"], c


def test_kswat_st(oracle, kat):
    for c in kat["kswat_st"]:
        got = oracle.kswat_st(c["q"], c["s"], c["qst"], c["sst"])
        assert got[0] == c["out"][0] and list(got[1:]) == c["out"][1:], (c, got)


def test_scalar_formulas(oracle, kat):
    for s, b in kat["score2bit"]:
        assert oracle.score2bit(s) == b
    for D, a, b, bit, e in kat["bit2e"]:
        assert oracle.bit2e(D, a, b, bit) == e
    for e, s in kat["f2s"]:
        assert oracle.f2s(float(e)) == s
    for x, s in kat["fmt_idy"]:
        assert oracle.fmt_idy(float(x)) == s


@pytest.mark.parametrize("name", golden_names())
def test_end_to_end_golden(oracle, name, tmp_path):
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    ref = os.path.join(GOLD, name + ".ref.fsa")
    qry = os.path.join(GOLD, name + ".qry.fsa") if meta["separate_query"] else ref
    out = str(tmp_path / "o.sc")
    subprocess.run([oracle.EXE, "-p", "blastp", "-i", qry, "-d", ref, "-o", out, "-T", str(tmp_path)] + meta["flags"], check=True,
                   stderr=subprocess.DEVNULL)
    assert open(out, "rb").read() == open(os.path.join(GOLD, name + ".sc"), "rb").read()


@pytest.mark.parametrize("name", het_golden_names())
def test_heterogeneous_lengths_golden(oracle, name, tmp_path):
    """300 proteins of log-normal length with subjects of 4562, 4597 and 30 014 residues: the REAL reference's rows for every
    query below 4096 residues (run in -l/-u ranges; it indexes an empty subject tile for longer queries, fsearch.py:1362/1396/
    1487-1490) against the oracle's, range by range."""
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    ref = os.path.join(GOLD, name + ".ref.fsa")
    got = b""
    for lo, hi in meta["ranges"]:
        out = str(tmp_path / ("r%d.sc" % lo))
        subprocess.run([oracle.EXE, "-p", "blastp", "-i", ref, "-d", ref, "-o", out, "-T", str(tmp_path), "-l", str(lo), "-u", str(hi)] + meta["flags"],
                       check=True, stderr=subprocess.DEVNULL)
        got += open(out, "rb").read()
    want = open(os.path.join(GOLD, name + ".sc"), "rb").read()
    assert want.count(b"\n") > 300 and meta["longest_subject"] >= 30000
    assert got == want


@pytest.mark.parametrize("name", ["stage_default", "stage_multi"])
def test_stage_dump(oracle, name):
    d = json.load(open(os.path.join(GOLD, name + ".stage.json")))
    fa = open(os.path.join(GOLD, name + ".ref.fsa"), "rb").read()
    ix = oracle.Index(fa, d["ssd"], d["nr"], d["step"], d["NC"])
    assert ix.threshold == d["threshold"]
    start, locus, soas = ix.start(), ix.locus(), ix.soas()
    assert len(locus) == d["n_locus"] and soas.tolist() == d["soas"]
    assert int(locus.astype(np.int64).sum()) == d["locus_sum"] and int(start.astype(np.int64).sum()) == d["start_sum"]
    assert locus[:4000].tolist() == d["locus_head"]
    nz = np.array(d["nonempty_buckets"])
    assert start[nz].tolist() == d["start_at_nonempty"]
    for q in d["queries"]:
        assert oracle.seg(q["masked"]) is not None
        assert ix.find_msav_m(q["masked"]) == q["cands"], q["i"]


def test_query_partition_invariance(oracle, tmp_path):
    """-l/-u splits concatenate to the full run (find_hit.py's block scheme relies on it)."""
    name = "toy_default"
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    ref = os.path.join(GOLD, name + ".ref.fsa")
    parts = b""
    for lo, hi in ((0, 40), (40, 99)):
        out = str(tmp_path / ("p%d.sc" % lo))
        subprocess.run([oracle.EXE, "-p", "blastp", "-i", ref, "-d", ref, "-o", out, "-l", str(lo), "-u", str(hi)] + meta["flags"],
                       check=True, stderr=subprocess.DEVNULL)
        parts += open(out, "rb").read()
    assert parts == open(os.path.join(GOLD, name + ".sc"), "rb").read()


def test_parallel_query_ranges_equal_one_call(oracle, tmp_path):
    """oracle.blastp_parallel (the GPU parity tests' oracle runs on many-core hosts: query ranges on threads, matrices per thread) gives
    the rows, records, candidate lists and counters of one call -- on a golden the REAL reference wrote, and on a synthetic set."""
    import numpy as np
    from swiftortho_amd import synthprot
    name = "toy_default"
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    ref = os.path.join(GOLD, name + ".ref.fsa")
    d = dict(zip(meta["flags"][0::2], meta["flags"][1::2]))
    kw = dict(ssd=d["-s"], nr=d["-r"], expect=float(d["-e"]), v=int(d["-v"]), step=int(d["-j"]), flt=d["-F"], ht=int(d["-M"]), chk=int(d["-c"]),
              max_miss=float(d.get("-m", 1e-3)), thr=int(d.get("-t", -1)))
    out = str(tmp_path / "g.sc")
    r = oracle.blastp_parallel(ref, ref, out, threads=3, min_piece=8, **kw)
    assert len(r.parts) == 3
    if int(d.get("-L", -1)) < 0 and int(d.get("-U", -1)) < 0:
        assert open(out, "rb").read() == open(os.path.join(GOLD, name + ".sc"), "rb").read()
    fa = str(tmp_path / "s.fsa")
    open(fa, "wb").write(synthprot.synthprot(500, 150, 3))
    kw = dict(ssd="111111", nr=oracle.AA9, expect=1e-5, v=500, step=1, flt="T", ht=1000003, chk=200)
    a = oracle.blastp(fa, fa, str(tmp_path / "a.sc"), st=30, ed=480, **kw)
    b = oracle.blastp_parallel(fa, fa, str(tmp_path / "b.sc"), threads=5, st=30, ed=480, **kw)
    assert len(b.parts) == 5
    assert open(str(tmp_path / "a.sc"), "rb").read() == open(str(tmp_path / "b.sc"), "rb").read()
    assert np.array_equal(a.ints, b.ints) and np.array_equal(a.dbl, b.dbl) and a.nqueries == b.nqueries == 450
    assert all(np.array_equal(a.cands(q), b.cands(q)) for q in range(a.nqueries))
    assert all(a.stats[k] == b.stats[k] for k in a.stats if not k.startswith("t_"))


def native_flags(p):
    """resolved launcher parameters -> the flags find_hit.py puts on the fsearch-c command line (find_hit.py:119-121)"""
    return ["-e", repr(p["exp"]), "-v", str(p["bv"]), "-L", str(p["rstart"]), "-U", str(p["rend"]), "-m", repr(p["miss"]), "-t", str(p["thr"]),
            "-j", str(p["step"]), "-F", p["flt"], "-M", str(p["ht"]), "-c", str(p["chk"]), "-s", p["ssd"], "-r", p["nr"]]


@pytest.mark.parametrize("name", launcher_golden_names())
def test_launcher_golden(oracle, name, tmp_path):
    """Host logic of the drop-in launcher (query_range, reference_parts, merge_parts) around the oracle's `fsearch-c`
    against outputs of the REAL bin/find_hit.py (tools/refharness/ref_find_hit.py)."""
    from swiftortho_amd import find_hit as fh
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    ref, qry = os.path.join(GOLD, name + ".ref.fsa"), os.path.join(GOLD, name + ".qry.fsa")
    p = fh.resolve(fh.parse(["find_hit.py", "-p", "blastp", "-i", qry, "-d", ref, "-o", "x"] + meta["find_hit_flags"]))
    nq = open(qry, "rb").read().count(b">")
    lo, hi = fh.query_range(p["start"], p["end"], nq, p["ngpu"])

    def native(ref_path, out):
        subprocess.run([oracle.EXE, "-p", "blastp", "-i", qry, "-d", ref_path, "-o", out, "-l", str(lo), "-u", str(hi)] + native_flags(p),
                       check=True, stderr=subprocess.DEVNULL)

    out = str(tmp_path / "o.sc")
    if meta["max_chr"] is None:
        native(ref, out)
    else:
        outs = []
        for k, text in enumerate(fh.reference_parts(ref, meta["max_chr"])):
            part = str(tmp_path / ("part%d.fsa" % k))
            open(part, "w", encoding="latin-1", newline="").write(text)
            outs.append(str(tmp_path / ("%d.sc" % k)))
            native(part, outs[-1])
        assert len(outs) > 2
        fh.merge_parts(sorted(outs), p["bv"], out)
    assert open(out, "rb").read() == open(os.path.join(GOLD, name + ".sc"), "rb").read()


def test_index_files_golden(oracle):
    """The reference's on-disk index files (tests/golden/idx_toy.*, written by the real Fasta.makedb / Fasta.write) against the
    oracle's chunk indexes: locus, soas, start and the parameter trailer, byte for byte."""
    meta = json.load(open(os.path.join(GOLD, "idx_toy.json")))
    a = meta["args"]
    fa = open(os.path.join(GOLD, "idx_toy.ref.fsa"), "rb").read()
    n = fa.count(b">")
    mw = max(sp.count("1") for sp in a["space"].split(","))
    for k, st in enumerate(range(0, n, a["chk"])):
        ed = min(st + a["chk"], n)
        ix = oracle.Index(fa, a["space"], a["nr"], a["step"], a["ht"], st, ed)
        assert ix.locus().astype("<i4").tobytes() == open(os.path.join(GOLD, "idx_toy.%d.idx" % k), "rb").read()
        assert ix.soas().astype("<i4").tobytes() == open(os.path.join(GOLD, "idx_toy.%d.soas" % k), "rb").read()
        trailer = "%d;%d;%d;%d;%d;%s;%s" % (st, ed + 1, mw, ix.threshold, a["ht"], a["space"], a["nr"])
        want = ix.start().astype("<i4").tobytes() + trailer.encode() + bytes([len(trailer)])
        assert want == open(os.path.join(GOLD, "idx_toy.%d.bin" % k), "rb").read()

"""BASELINE config 5: find_hit -> find_orth -> find_cluster -a mcl -I 1.5 end to end, ortholog groups compared with the
reference pipeline's (tools/refharness/make_pipeline_golden.py: oracle search rows, then the REAL bin/find_orth.py and
bin/find_cluster.py).  The proteome is regenerated from synthprot (md5-checked), so the fixtures are digests + groups."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

from conftest import GOLD, ROOT


def canonical(groups_text):
    rows = sorted("\t".join(sorted(l.split("\t"))) for l in groups_text.split("\n") if l)
    return "\n".join(rows) + "\n"


def downstream(sc_path, meta, tmp_path, cpu_oracle_mcl=False):
    """the two drop-in CLIs on an .sc file -> (orth text, groups text).  bin/find_cluster.py runs its Markov loop on the GPU; on the
    CPU side of the suite the same host code is called in-process with the scipy oracle of that loop plugged in."""
    orth = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "find_orth.py"), "-i", sc_path] + meta["find_orth_flags"], capture_output=True, check=True).stdout
    op = str(tmp_path / "x.orth")
    open(op, "wb").write(orth)
    if cpu_oracle_mcl:
        from mcl_scipy_oracle import scipy_mcl
        from swiftortho_amd import find_cluster as fc
        a = fc.parse(["find_cluster.py", "-i", op] + meta["find_cluster_flags"])
        with open(op) as f:
            groups = "".join("\t".join(g) + "\n" for g in fc.cnc(f, float(a["-I"]), mcl=scipy_mcl))
        return orth, groups
    groups = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "find_cluster.py"), "-i", op] + meta["find_cluster_flags"], capture_output=True,
                            check=True).stdout
    return orth, groups.decode()


def check(meta, orth, groups, name):
    assert hashlib.md5(orth).hexdigest() == meta["orth_md5"], "find_orth output differs from the reference's"
    assert groups.count("\n") == meta["groups"] and len(groups.split()) == meta["genes_in_groups"]
    assert hashlib.sha256(canonical(groups).encode()).hexdigest() == meta["groups_canonical_sha256"], "ortholog groups differ (as sets)"
    assert hashlib.sha256(groups.encode()).hexdigest() == meta["groups_text_sha256"]
    want = open(os.path.join(GOLD, name + ".groups")).read()
    assert groups.startswith(want)      # c2: the whole file; c3: its first 200 groups


def test_downstream_of_oracle_search_c2(oracle, tmp_path):
    """CPU: oracle rows of the 10k-protein search -> find_orth -> find_cluster == the reference pipeline"""
    from swiftortho_amd import synthprot
    meta = json.load(open(os.path.join(GOLD, "pipe_c2.json")))
    fa = synthprot.synthprot(meta["proteins"], 300)
    assert hashlib.md5(fa).hexdigest() == meta["fasta_md5"]
    p = str(tmp_path / "x.fsa")
    open(p, "wb").write(fa)
    P, n = os.cpu_count() or 1, meta["proteins"]
    block = (n + P - 1) // P
    procs = [subprocess.Popen([oracle.EXE, "-p", "blastp", "-i", p, "-d", p, "-o", str(tmp_path / ("%03d.sc" % k)), "-l", str(k * block), "-u",
                               str(min(n, (k + 1) * block))] + meta["find_hit_flags"], stderr=subprocess.DEVNULL) for k in range(P)]
    assert all(q.wait() == 0 for q in procs)
    sc = str(tmp_path / "all.sc")
    with open(sc, "wb") as o:
        for k in range(P):
            o.write(open(str(tmp_path / ("%03d.sc" % k)), "rb").read())
    assert hashlib.md5(open(sc, "rb").read()).hexdigest() == meta["sc_md5"]
    orth, groups = downstream(sc, meta, tmp_path, cpu_oracle_mcl=True)
    check(meta, orth, groups, "pipe_c2")


def launcher_flags(native):
    """fsearch-c flags of the golden -> bin/find_hit.py flags (same letters; the alphabet by name)"""
    d = dict(zip(native[0::2], native[1::2]))
    return ["-e", d["-e"], "-v", d["-v"], "-j", d["-j"], "-F", d["-F"], "-s", d["-s"], "-r", "aa9", "-M", d["-M"], "-c", d["-c"]]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["c2", "c3"])
def test_pipeline_on_gpu(cfg, tmp_path):
    """GPU: bin/find_hit.py -> bin/find_orth.py -> bin/find_cluster.py; c3 = BASELINE config 5 at full size (100k proteins)"""
    from swiftortho_amd import synthprot
    path = os.path.join(GOLD, "pipe_%s.json" % cfg)
    if not os.path.isfile(path):
        pytest.skip("no golden for " + cfg)
    meta = json.load(open(path))
    fa = synthprot.synthprot(meta["proteins"], 300)
    assert hashlib.md5(fa).hexdigest() == meta["fasta_md5"]
    p = str(tmp_path / "x.fsa")
    open(p, "wb").write(fa)
    sc = str(tmp_path / "x.sc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "find_hit.py"), "-p", "blastp", "-i", p, "-d", p, "-o", sc, "-a", "1"] + launcher_flags(meta["find_hit_flags"]),
                       capture_output=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert hashlib.md5(open(sc, "rb").read()).hexdigest() == meta["sc_md5"], "search rows differ from the oracle's"
    orth, groups = downstream(sc, meta, tmp_path)
    check(meta, orth, groups, "pipe_" + cfg)


@pytest.mark.gpu
def test_orthology_from_hit_records_on_gpu(tmp_path):
    """the in-process flow (swiftortho_amd/pipeline.py): GPU search, then find_orth on the hit RECORDS -- no .sc is parsed -- gives
    the reference pipeline's relations byte for byte; the .sc it can still write is the oracle's"""
    from swiftortho_amd import pipeline, synthprot
    meta = json.load(open(os.path.join(GOLD, "pipe_c2.json")))
    fa = synthprot.synthprot(meta["proteins"], 300)
    p, sc = str(tmp_path / "x.fsa"), str(tmp_path / "x.sc")
    open(p, "wb").write(fa)
    d = dict(zip(meta["find_hit_flags"][0::2], meta["find_hit_flags"][1::2]))
    lines, times = pipeline.orthology_from_search(p, sc_path=sc, ssd=d["-s"], nr=d["-r"], ht=int(d["-M"]), chk=int(d["-c"]), step=int(d["-j"]), v=int(d["-v"]),
                                                  expect=float(d["-e"]), flt=d["-F"])
    assert times["rows"] == meta["sc_rows"]
    assert hashlib.md5(open(sc, "rb").read()).hexdigest() == meta["sc_md5"]
    assert hashlib.md5(b"".join(l + b"\n" for l in lines)).hexdigest() == meta["orth_md5"]


@pytest.mark.gpu
def test_run_all_fast_wrapper_clusters_the_recoded_file(tmp_path):
    """scripts/run_all_fast.py's clustering step as the reference wrapper has it (run_all_fast.py:139-193): ids recoded to numbers by
    first appearance (.xyz), THAT file clustered, numbers mapped back (.clsr), .grp removed.  (For -A mcl the reference calls the
    external mcl program; here bin/find_cluster.py -a mcl clusters the same .xyz: consistency of the wrapper's own files is what
    this checks -- the docstring of the script says what is not pinned.)"""
    from swiftortho_amd import synthprot
    fas = str(tmp_path / "w.fsa")
    open(fas, "wb").write(synthprot.synthprot(1500, 150, 21))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_all_fast.py"), "-i", fas, "-s", "111111", "-a", "1", "-v", "500"],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    res = fas + "_results"
    opc = [l.split("\t") for l in open(os.path.join(res, "w.fsa.opc"))]
    xyz = [l.split("\t") for l in open(os.path.join(res, "w.fsa.xyz"))]
    assert len(opc) == len(xyz) > 100 and not os.path.exists(os.path.join(res, "w.fsa.grp"))
    id2n = {}
    for (typ, q, s, sco), (a, b, z) in zip(opc, xyz):
        for g in (q, s):
            id2n.setdefault(g, len(id2n))
        assert (a, b, z) == (str(id2n[q]), str(id2n[s]), sco)
    grp = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "find_cluster.py"), "-i", os.path.join(res, "w.fsa.xyz"), "-a", "mcl", "-I", "1.5"],
                         capture_output=True, text=True, check=True).stdout
    n2id = {str(n): g for g, n in id2n.items()}
    want = "".join("\t".join(n2id[k] for k in l.split("\t")) + "\n" for l in grp.split("\n") if l)
    got = open(os.path.join(res, "w.fsa.clsr")).read()
    assert got == want and got.count("\n") > 20
    genes = got.split()
    assert len(genes) == len(set(genes)) and set(genes) <= set(id2n)

"""GPU parity tests: libsohit.so (HIP, through the C ABI) against the oracle and the golden
fixtures.  Bit-exact for every integer/byte/index column; e-value and identity are compared as
exact doubles too (both sides evaluate the same IEEE expressions), which is tighter than the
1e-6 relative tolerance BASELINE.json allows.

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLD, golden_names, launcher_golden_names

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6  # tolerance BASELINE.json states for e-value / bit-score columns


def flags_to_kwargs(flags):
    d = dict(zip(flags[0::2], flags[1::2]))
    nr = d.get("-r", "AST,CFILMVY,DN,EQ,G,H,KR,P,W")
    return dict(ssd=d.get("-s", "111111"), nr=nr, ht=int(d.get("-M", -1)), chk=int(d.get("-c", 50000)), step=int(d.get("-j", 4)),
                v=int(d.get("-v", 500)), thr=int(d.get("-t", -1)), expect=float(d.get("-e", 1e-3)), max_miss=float(d.get("-m", 1e-3)),
                flt=d.get("-F", "T"))


@pytest.fixture(scope="module")
def fs():
    from swiftortho_amd import fsearch
    return fsearch


def gpu_rows(fs, ref_bytes, qry_bytes, kw, st=-1, ed=-1, keep=False):
    if keep:
        os.environ["SOHIT_KEEP_CANDS"] = "1"
        os.environ["SOHIT_KEEP_MASKED"] = "1"
    s = fs.Searcher(**kw)
    s.load_ref_bytes(ref_bytes)
    s.load_queries_bytes(qry_bytes)
    hits = s.search(st, ed)
    rows = b"".join(hits.rows())
    return s, hits, rows


@pytest.mark.parametrize("name", golden_names())
def test_end_to_end_golden(fs, name):
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    ref = open(os.path.join(GOLD, name + ".ref.fsa"), "rb").read()
    qry = open(os.path.join(GOLD, name + ".qry.fsa"), "rb").read() if meta["separate_query"] else ref
    s, hits, rows = gpu_rows(fs, ref, qry, flags_to_kwargs(meta["flags"]))
    want = open(os.path.join(GOLD, name + ".sc"), "rb").read()
    if rows != want:
        a, b = rows.split(b"\n"), want.split(b"\n")
        for i in range(max(len(a), len(b))):
            x = a[i] if i < len(a) else b"<none>"
            y = b[i] if i < len(b) else b"<none>"
            assert x == y, "row %d differs\n gpu: %s\n ref: %s" % (i, x.decode("latin-1"), y.decode("latin-1"))
    hits.close()
    s.close()


@pytest.mark.parametrize("name", ["stage_default", "stage_multi"])
def test_index_and_candidates_vs_golden(fs, oracle, name):
    d = json.load(open(os.path.join(GOLD, name + ".stage.json")))
    fa = open(os.path.join(GOLD, name + ".ref.fsa"), "rb").read()
    kw = dict(ssd=d["ssd"], nr=d["nr"], ht=d["NC"], chk=50000, step=d["step"], v=500, expect=1e-5, flt="T")
    s, hits, _ = gpu_rows(fs, fa, fa, kw, keep=True)
    assert s.chunk_threshold(0) == d["threshold"]
    ix = oracle.Index(fa, d["ssd"], d["nr"], d["step"], d["NC"])
    check_index(s, ix, d["NC"], A=d["nr"].count("/") + 1, S=d["ssd"].count(",") + 1)
    for q in d["queries"]:
        assert s.masked_query(q["i"]).decode("latin-1") == q["masked"]
        got = s.query_candidates(q["i"]).astype(np.int64).tolist()
        assert got == q["cands"], "query %d" % q["i"]
    hits.close()
    s.close()


def check_index(s, ix, NC, A, S):
    """GPU chunk index == oracle CSR: same bucket boundaries, and the downloaded entries in the reference's slot order (descending entry
    value inside a bucket: so_chunk_download puts a built chunk in that order, as the first dense pass would)."""
    o_start, o_locus, soas = ix.start(), ix.locus(), ix.soas()
    g_start, ent = s.chunk_index(0)
    E = len(o_locus)
    assert int(g_start[NC]) == E and len(ent) == E
    assert np.array_equal(g_start[:NC], o_start)
    if E == 0:
        return
    subj = (ent >> np.uint64(32)).astype(np.int64)
    pos = (ent & np.uint64(0xFFFFFF)).astype(np.int64)
    x = soas[subj].astype(np.int64) + pos
    counts = np.diff(g_start.astype(np.int64))
    bucket = np.repeat(np.arange(NC), counts)
    # reference order inside a bucket = descending entry value (subject, tag, pos)
    order = np.lexsort(((~ent), bucket))
    assert np.array_equal(x[order], o_locus.astype(np.int64))
    assert np.array_equal(order, np.arange(E)), "entries are not in the reference's slot order"
    # the slot the reference never reads
    last_b = bucket[-1]
    lo = int(g_start[last_b])
    assert ent[E - 1] == ent[lo:E].min()


# The oracle's answer for one (input, parameters, range) is computed once per test session: the cases that search the same set under several
# switch settings (the oracle has no switches) share it.  Key: digest of the FASTA bytes + every oracle argument.
_ORACLE_RUNS = {}


def oracle_run(oracle, fasta, kw, st, ed, tmp_path):
    import hashlib
    key = (hashlib.sha1(fasta).hexdigest(), kw["ssd"], kw["nr"], kw["expect"], kw["v"], kw["step"], kw["flt"], kw["ht"], kw["chk"], st, ed,
           kw.get("thr", -1), kw.get("max_miss", 1e-3))
    hit = _ORACLE_RUNS.get(key)
    if hit is None:
        fa = str(tmp_path / "x.fsa")
        open(fa, "wb").write(fasta)
        out = str(tmp_path / "o.oracle.sc")
        # (query ranges side by side on the host's cores: identical to one call, tests/test_oracle_golden.py checks that)
        r = oracle.blastp_parallel(fa, fa, out, ssd=kw["ssd"], nr=kw["nr"], expect=kw["expect"], v=kw["v"], step=kw["step"], flt=kw["flt"],
                                   ht=kw["ht"], chk=kw["chk"], st=st, ed=ed, thr=kw.get("thr", -1), max_miss=kw.get("max_miss", 1e-3))
        hit = _ORACLE_RUNS[key] = (r, open(out, "rb").read())
    return hit


def oracle_vs_gpu(fs, oracle, fasta, kw, tmp_path, sub=None):
    st, ed = sub if sub else (-1, -1)
    r, want = oracle_run(oracle, fasta, kw, st, ed, tmp_path)
    open(str(tmp_path / "o.sc"), "wb").write(want)   # (some callers read the oracle's text back)
    s, hits, rows = gpu_rows(fs, fasta, fasta, kw, st, ed, keep=True)
    # stage: candidates of every query, in the reference's spill order
    lo = 0 if st < 0 else st
    for qrel in range(r.nqueries):
        got = s.query_candidates(lo + qrel)
        exp = r.cands(qrel)
        assert np.array_equal(got, exp), "candidates of query %d differ:\n gpu %s\n ora %s" % (lo + qrel, got[:8], exp[:8])
    a, b = rows.split(b"\n"), want.split(b"\n")
    for i in range(max(len(a), len(b))):
        x = a[i] if i < len(a) else b"<none>"
        y = b[i] if i < len(b) else b"<none>"
        assert x == y, "row %d differs\n gpu: %s\n ora: %s" % (i, x.decode("latin-1"), y.decode("latin-1"))
    # fixed-width records: integer columns exact, fp columns within the stated tolerance
    g = hits.array()
    assert len(g) == len(r.ints)
    if len(g):
        for k, col in enumerate(["qidx", "sidx", "aln", "mis", "gap", "qst", "qed", "sst", "sed", "bit", "qlen", "slen", "ungapped"]):
            assert np.array_equal(g[col].astype(np.int64), r.ints[:, k]), col
        assert np.allclose(g["identity"], r.dbl[:, 0], rtol=REL_TOL, atol=0)
        assert np.allclose(g["evalue"], r.dbl[:, 1], rtol=REL_TOL, atol=0)
        assert np.array_equal(g["identity"], r.dbl[:, 0]) and np.array_equal(g["evalue"], r.dbl[:, 1])
    c = s.counters()
    assert c["seed_hits"] == r.stats["seed_hits"]
    if os.environ.get("SOHIT_UG_COUNT") == "1":
        # the extension kernels' counting instances: b62 lookups == the reference's `flag` sum (fsearch.py:2467, 2482) -- the EXTENTS of every
        # ungapped pass are the reference's, not only the scores that reach a candidate
        assert c["ungap_steps"] == r.stats["ungap_steps"], (c["ungap_steps"], r.stats["ungap_steps"])
    # the oracle also counts the never-scoring "subject -1" groups (offset 0 of a chunk's first sequence)
    assert 0 <= r.stats["groups"] - c["groups"] <= 64 * c["n_chunks"] * max(1, r.nqueries // 50)
    hits.close()
    s.close()
    return c, r.stats


@pytest.mark.parametrize("env", [{}, {"SOHIT_BUCKET_MIN": "0"}, {"SOHIT_BUCKET_MIN": "0", "SOHIT_UG1": "0"}, {"SOHIT_BUCKET_MIN": "0", "SOHIT_UG1_CHAIN": "0"},
                                 {"SOHIT_BUCKET_MIN": "0", "SOHIT_UG_W32": "0"}],
                         ids=["default", "bucketed_ungap1_ungap2", "bucketed_classic", "bucketed_ungap1_classic_chains", "bucketed_keys"])
def test_ungap_steps_equal_the_reference_count(fs, oracle, tmp_path, monkeypatch, env):
    """so_counters.ungap_steps (SOHIT_UG_COUNT=1: counting instances of k_ungap / k_ungap1 / k_ungap2) == the oracle's sum of Fasta.ungap's
    `flag` over all groups, on homologous families (chains), a uniform set (singletons) and mixed lengths (banded diagonals), whichever
    kernels the pass takes; the counters say which ones ran."""
    from swiftortho_amd import synthprot
    monkeypatch.setenv("SOHIT_UG_COUNT", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    for fa in (synthprot.synthprot(1500, 300, 91), synthprot.uniform_proteins(900, 250, 92), synthprot.synthprot(700, seed=93, lengths="lognormal")):
        c, st = oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)
        assert c["ungap_steps"] > 10 * c["groups"]
        if "SOHIT_UG1" in env or "SOHIT_UG_W32" in env:
            assert c["groups_single"] == 0 and c["groups_chain"] == 0
        elif env.get("SOHIT_BUCKET_MIN") == "0":
            assert c["groups_single"] > 0.5 * c["groups"]
            assert (c["groups_chain"] > 0) == ("SOHIT_UG1_CHAIN" not in env)


@pytest.mark.parametrize("uq", ["1", "0"], ids=["wave_per_query", "sorted_keys_only"])
def test_sparse_pass_singletons_vs_oracle(fs, oracle, tmp_path, monkeypatch, uq):
    """A sparse pass (long seed: a few hundred hits per query and chunk) takes the sorted path; the hits that are alone on their diagonal
    are found and extended by k_ungapq, a wave per query, without a key (default), or every hit goes through the sorted keys
    (SOHIT_UNGAPQ=0).  Rows, candidates and the number of BLOSUM lookups are the oracle's both ways, on uniform and mixed lengths, and
    the counters say which kernel took the singletons."""
    from swiftortho_amd import synthprot
    monkeypatch.setenv("SOHIT_UG_COUNT", "1")
    monkeypatch.setenv("SOHIT_UNGAPQ", uq)
    kw = dict(ssd="11111011111", nr=oracle.AA9, ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    c, st = oracle_vs_gpu(fs, oracle, synthprot.synthprot(1500, 300, 91), kw, tmp_path)
    assert c["hits_bucketed"] == 0 and c["groups_chain"] == 0
    assert (c["groups_single"] > 0.3 * c["groups"]) if uq == "1" else c["groups_single"] == 0
    # mixed lengths: the classes above 512 residues (and the passes that merge them) keep the sorted keys for every hit
    oracle_vs_gpu(fs, oracle, synthprot.synthprot(700, seed=93, lengths="lognormal"), kw, tmp_path)


@pytest.mark.parametrize("lazy", ["1", "0"], ids=["orders_on_demand", "orders_with_the_batch"])
def test_kmer_orders_on_demand_vs_oracle(fs, oracle, tmp_path, monkeypatch, lazy):
    """The frequency cap (fsearch.py:2667-2677) walks a query's windows in k-mer score order, but a query whose windows together stay
    below the limit keeps them all whatever the order (k_cap_all): orders are computed for the queries that reach their cap, in the
    chunk where they do -- or for every query at the first chunk where such queries are the majority.  Chunk 0 holds iid proteins (a
    query meets itself at most: below the cap of 1 x length), family members in chunks 1 and 2 exceed it: rows and candidates are the
    oracle's whether the orders come on demand (default) or with the batch (SOHIT_KSC_LAZY=0), and the profile keys say which chunks
    asked.  A set in which every protein occurs twice has every query over its cap; without a cap in reach nothing is ordered; a
    mixed-length set has open queries whose order needs global scratch (side stream, their class's cap deferred)."""
    from swiftortho_amd import synthprot
    monkeypatch.setenv("SOHIT_KSC_LAZY", lazy)

    def asked(fa, kw):
        s, hits, _ = gpu_rows(fs, fa, fa, kw)
        hits.close()
        s.set_profile(True)
        s.search(0, -1).close()
        t = s.timing()
        s.close()
        return int(t.get("seed.kmer_orders_open_chunks", 0)), int(t.get("seed.kmer_orders_all_at_chunk", 0))

    fa = synthprot.uniform_proteins(400, 200, 96).replace(b"|p", b"|u") + synthprot.synthprot(800, 200, 97)
    kw = dict(ssd="11111011111", nr=oracle.AA9, ht=120000000, chk=400, step=1, v=500, expect=1e-5, flt="T", thr=1)
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)
    some, everybody = asked(fa, kw)
    if lazy == "1":
        assert some & 1 == 0 and some & 6 != 0 and everybody == 0, (some, everybody)
    else:
        assert (some, everybody) == (0, 0)
    assert asked(fa, dict(kw, thr=100000)) == (0, 0)
    twice = synthprot.synthprot(300, 200, 98)
    twice = twice + twice.replace(b"|p", b"|d")
    kw1 = dict(kw, chk=50000)   # (one chunk: both copies in it)
    oracle_vs_gpu(fs, oracle, twice, kw1, tmp_path)
    assert asked(twice, kw1) == ((0, 1) if lazy == "1" else (0, 0))
    het = synthprot.synthprot(1200, seed=99, lengths="lognormal")
    oracle_vs_gpu(fs, oracle, het, dict(kw1, chk=500), tmp_path)
    if lazy == "1":
        assert asked(het, dict(kw1, chk=500))[0] != 0


def test_synth_2000_vs_oracle(fs, oracle, tmp_path):
    from swiftortho_amd import synthprot
    kw = dict(ssd="111111", nr=oracle.AA9, ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    oracle_vs_gpu(fs, oracle, synthprot.synthprot(2000, 300, 77), kw, tmp_path)


def test_synth_multichunk_vs_oracle(fs, oracle, tmp_path):
    from swiftortho_amd import synthprot
    kw = dict(ssd="111111", nr=oracle.AA9, ht=5000011, chk=700, step=1, v=500, expect=1e-5, flt="T")
    oracle_vs_gpu(fs, oracle, synthprot.synthprot(1500, 200, 78), kw, tmp_path)


def test_uniform_adversarial_vs_oracle(fs, oracle, tmp_path):
    """iid residues, no homologs: the seed cap binds and the early-stop rule fires."""
    from swiftortho_amd import synthprot
    kw = dict(ssd="111111", nr=oracle.AA9, ht=15000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    oracle_vs_gpu(fs, oracle, synthprot.uniform_proteins(1200, 300, 79), kw, tmp_path)


def test_weight10_seed_vs_oracle(fs, oracle, tmp_path):
    from swiftortho_amd import synthprot
    kw = dict(ssd="11111011111", nr=oracle.AA9, ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    oracle_vs_gpu(fs, oracle, synthprot.synthprot(3000, 300, 80), kw, tmp_path)


def test_multiseed_two_alphabets_vs_oracle(fs, oracle, tmp_path):
    from swiftortho_amd import synthprot
    kw = dict(ssd="111111,1101011", nr=oracle.AA9 + "/A,KR,EDNQ,C,G,H,ILVM,FYW,P,ST", ht=3000017, chk=400, step=2, v=50,
              expect=1e-3, flt="T")
    oracle_vs_gpu(fs, oracle, synthprot.synthprot(900, 150, 81), kw, tmp_path)


def test_long_sequences_self_search_vs_oracle(fs, oracle, tmp_path):
    """>= 4096-aa proteins in a self-search: tiled long path (kswat_st_long) for long queries and
    long subjects, including tiles that run past a shorter subject (undefined in the reference,
    defined here and in the oracle as "no hit from that tile")."""
    from swiftortho_amd import synthprot
    rng = np.random.default_rng(5)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)

    def rnd(n):
        return aa[rng.integers(0, 20, n)].tobytes().decode()

    def mut(s, d):
        b = np.frombuffer(s.encode(), dtype=np.uint8).copy()
        m = rng.random(len(b)) < d
        b[m] = aa[rng.integers(0, 20, int(m.sum()))]
        return b.tobytes().decode()

    A, B = rnd(9500), rnd(4300)
    recs = [("L0", A), ("L1", mut(A[:6200], 0.15)), ("L2", mut(A[2500:7000], 0.3)), ("L3", B), ("L4", mut(B, 0.2) + rnd(300)),
            ("S0", mut(A[100:420], 0.1)), ("S1", mut(A[5000:5350], 0.2)), ("S2", mut(A[9000:9490], 0.1)), ("S3", mut(B[3900:4290], 0.1))]
    fa = "".join(">%s\n%s\n" % r for r in recs).encode() + synthprot.synthprot(300, 250, 46)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=1000003, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)


def test_query_subrange_and_small_batches(fs, oracle, tmp_path, monkeypatch):
    """-l/-u sub-range, processed in several device batches: rows must not depend on batching."""
    from swiftortho_amd import synthprot
    monkeypatch.setenv("SOHIT_BATCH", "64")
    kw = dict(ssd="111111", nr=oracle.AA9, ht=1000003, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    oracle_vs_gpu(fs, oracle, synthprot.synthprot(500, 120, 82), kw, tmp_path, sub=(100, 333))


def test_full_candidate_store_splits_the_batch(fs, oracle, tmp_path, monkeypatch):
    """A batch whose 32-bit candidate store would overflow is run again as two halves (host_search.hip run_batch).  SOHIT_CAND_LIMIT lowers
    the limit from 2^32 so that the first attempt -- and the first halves -- overflow: rows, candidates and counters are unchanged;
    a limit below one query's own candidates is an error, not a loop."""
    from swiftortho_amd import synthprot
    fa = synthprot.synthprot(500, 120, 82)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=1000003, chk=200, step=1, v=500, expect=1e-5, flt="T")
    c0, _ = oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)
    assert c0["n_chunks"] >= 2 and c0["candidates"] > 2000
    monkeypatch.setenv("SOHIT_CAND_LIMIT", str(c0["candidates"] // 5))
    c1, _ = oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)
    for k in ("candidates", "seed_hits", "groups", "rows", "n_queries", "query_aa", "alignments"):
        assert c1[k] == c0[k], k
    monkeypatch.setenv("SOHIT_CAND_LIMIT", "1")
    with pytest.raises(Exception, match="one query collected"):
        gpu_rows(fs, fa, fa, kw, -1, -1)


@pytest.mark.parametrize("parts", [3, 8])
def test_rows_leave_in_query_ranges(fs, oracle, tmp_path, monkeypatch, parts):
    """Large results are traced, written and downloaded in several query ranges (host_phase2.hip phase2, SOHIT_EMIT_PARTS); forced here on a
    small input (SOHIT_EMIT_MIN_ROWS=1), with more ranges than some batches have queries, several batches, and queries without rows."""
    from swiftortho_amd import synthprot
    monkeypatch.setenv("SOHIT_EMIT_PARTS", str(parts))
    monkeypatch.setenv("SOHIT_EMIT_MIN_ROWS", "1")
    kw = dict(ssd="111111", nr=oracle.AA9, ht=1000003, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    fa = synthprot.synthprot(700, 150, 311) + b">lonely\n" + synthprot.uniform_proteins(1, 200, 5).split(b"\n", 1)[1]
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)
    monkeypatch.setenv("SOHIT_BATCH", "5")
    oracle_vs_gpu(fs, oracle, synthprot.synthprot(60, 120, 17), kw, tmp_path, sub=(3, 41))


@pytest.mark.parametrize("env", [{"SOHIT_SPEC": "0"}, {"SOHIT_SPEC": "1", "SOHIT_SPEC_SLACK": "1e30"}, {"SOHIT_SPEC": "1", "SOHIT_SPEC_SLACK": "1e-30"},
                                 {"SOHIT_SPEC": "1", "SOHIT_SPEC_CAP": "50"}, {"SOHIT_SPEC": "1"},
                                 {"SOHIT_SPEC": "1", "SOHIT_SPEC_SLACK": "1e4", "SOHIT_EMIT_MIN_ROWS": "1"}])
def test_speculative_traces_do_not_change_rows(fs, oracle, tmp_path, monkeypatch, env):
    """First-round tasks whose ungapped score alone would pass the e-value test are aligned WITH traces at once (host_phase2.hip phase2,
    k_round_counts_spec); reported rows that have a trace skip the second alignment.  Rows are the oracle's with the guess switched
    off, with every first-round task traced, with none, with a trace budget the first round does not fit (the round is redone without
    traces), and together with the ranged row emission; long candidates (tiled alignments) included."""
    from swiftortho_amd import synthprot
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=1000003, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    oracle_vs_gpu(fs, oracle, synthprot.synthprot(600, 160, 411), kw, tmp_path)
    kw2 = dict(kw, v=3, expect=1e-3)
    oracle_vs_gpu(fs, oracle, synthprot.synthprot(40, 4500, 7), kw2, tmp_path)


def test_aligner_launch_order_and_result_cache_do_not_change_rows(fs, oracle, tmp_path, monkeypatch):
    """The score-only aligner launches are ordered by band rows (k_task_rows + radix sort, lists of >= 4096 tasks) and a released
    result array is reused by the next search (so_free_hits keeps one): with both switched off, and over repeated searches on one
    context (the second search starts from the first one's array), rows stay the oracle's."""
    from swiftortho_amd import synthprot
    fa = synthprot.synthprot(3000, 220, 97)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=4000037, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)
    want = open(str(tmp_path / "o.sc"), "rb").read()
    big = synthprot.synthprot(12000, 300, 98)   # ~2 MB of result records: above the cache's 1 MiB floor
    s = fs.Searcher(**kw)
    s.load_ref_bytes(big)
    s.load_queries_bytes(big)
    first = None
    for _ in range(3):
        h = s.search()
        rows = b"".join(h.rows())
        assert len(h) * 80 > (1 << 20)
        first = first or rows
        assert rows == first
        h.close()
    s.load_ref_bytes(fa)          # a smaller result written into the kept (larger) array
    s.load_queries_bytes(fa)
    h = s.search()
    assert b"".join(h.rows()) == want
    h.close()
    s.close()
    monkeypatch.setenv("SOHIT_ALIGN_SORT", "0")
    monkeypatch.setenv("SOHIT_HIT_CACHE", "0")
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)


def test_bucket_directory_map_fallback(fs, oracle, tmp_path, monkeypatch):
    """The chunk's bucket directory is a bitmap + rank table up to -M 2^28 and an open-addressed map above (SOHIT_DIR_MAX moves
    the limit): the map path, forced here, gives the same rows, candidates and counters; three chunks, colliding buckets."""
    from swiftortho_amd import synthprot
    monkeypatch.setenv("SOHIT_DIR_MAX", "0")
    kw = dict(ssd="111111", nr=oracle.AA9, ht=50021, chk=300, step=1, v=500, expect=1e-5, flt="T")
    oracle_vs_gpu(fs, oracle, synthprot.synthprot(800, 200, 99), kw, tmp_path)


def test_no_filter_threshold_override_small_v(fs, oracle, tmp_path):
    """-F F (no SEG masking), -t override of the seed-frequency threshold, -v 3, -m 0.5, lower-case residues"""
    from swiftortho_amd import synthprot
    fa = synthprot.synthprot(600, 180, 85)
    lines = fa.split(b"\n")
    for k in range(1, len(lines), 14):   # lower-case every 7th sequence: hashed/scored like upper case, never identical to it
        lines[k] = lines[k].lower()
    fa = b"\n".join(lines)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=2000003, chk=50000, step=1, v=3, expect=1e-3, flt="F", thr=7, max_miss=0.5)
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)


def test_hit_budget_splits_passes(fs, oracle, tmp_path, monkeypatch):
    """a tiny per-pass hit budget forces many query sub-range passes per (batch, chunk)"""
    from swiftortho_amd import synthprot
    monkeypatch.setenv("SOHIT_MAX_HITS", "20000")
    kw = dict(ssd="111111", nr=oracle.AA9, ht=1000003, chk=150, step=1, v=500, expect=1e-5, flt="T")
    oracle_vs_gpu(fs, oracle, synthprot.synthprot(400, 150, 84), kw, tmp_path)


@pytest.mark.parametrize("NC", [9, 16384, 16385, 134217728, 134217729, 300000007])
def test_index_grouping_at_every_bucket_count(fs, oracle, NC):
    """The index build groups its (bucket, entry) pairs with two hand-written counting passes (k_ixsort.hip): bins of 2^wsh ids, then
    16384 ids per LDS pass -- one bin (NC <= 16384), the first two-bin split, the widest single-pass bins (NC = 2^27) and bins that
    need the extra split by the upper digit (NC above 2^27; 3e8 also takes the hashed directory).  Index arrays == the oracle's CSR."""
    from swiftortho_amd import synthprot
    fa = synthprot.synthprot(300, 120, 17)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=NC, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    s, hits, _ = gpu_rows(fs, fa, fa, kw)
    ix = oracle.Index(fa, kw["ssd"], kw["nr"], 1, NC)
    assert s.chunk_threshold(0) == ix.threshold
    check_index(s, ix, NC, A=1, S=1)
    hits.close()
    s.close()


def test_huge_family_over_4096_candidates_per_query(fs, oracle, tmp_path):
    """one 6000-member family: every query collects > 4096 candidates, which takes the global-memory
    variant of the exact wave quicksort (phase 2 ordering) instead of the LDS one."""
    rng = np.random.default_rng(11)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    anc = aa[rng.integers(0, 20, 150)]
    recs = []
    for i in range(6000):
        b = anc.copy()
        m = rng.random(150) < rng.uniform(0.05, 0.3)
        b[m] = aa[rng.integers(0, 20, int(m.sum()))]
        recs.append(">f%d\n%s\n" % (i, b.tobytes().decode()))
    kw = dict(ssd="111111", nr=oracle.AA9, ht=3000017, chk=50000, step=1, v=500, expect=1e-5, flt="T", thr=100000)
    c, _ = oracle_vs_gpu(fs, oracle, "".join(recs).encode(), kw, tmp_path, sub=(17, 23))
    assert c["candidates"] > 6 * 4096
    # -v 2500: every query reports > 1024 rows, so the final selection takes the one-thread quicksort replay instead of the
    # LDS wave one (k_final_select_lds serves lists of up to 1024 rows)
    kw["v"] = 2500
    c, _ = oracle_vs_gpu(fs, oracle, "".join(recs).encode(), kw, tmp_path, sub=(40, 43))
    assert c["rows"] > 3 * 1024


def test_wide_addends_and_device_wide_sort_paths(fs, oracle, tmp_path, monkeypatch):
    """the 8-byte key-addend lookup path (taken when subject + diagonal + tag fields exceed 32 bits) and the
    device-wide key sort (taken for passes with few queries), forced on an ordinary input"""
    from swiftortho_amd import synthprot
    monkeypatch.setenv("SOHIT_LK_WIDE", "1")
    monkeypatch.setenv("SOHIT_SEGSORT", "0")
    kw = dict(ssd="111111,1101011", nr=oracle.AA9, ht=2000003, chk=900, step=1, v=500, expect=1e-5, flt="T")
    oracle_vs_gpu(fs, oracle, synthprot.synthprot(1300, 200, 86), kw, tmp_path)


def test_very_long_sequences_40k(fs, oracle, tmp_path):
    """40 000-residue proteins (ten 4096-tiles per alignment, 16-bit position fields, SEG tiles) and a
    low-complexity repeat next to ordinary sequences"""
    from swiftortho_amd import synthprot
    rng = np.random.default_rng(3)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)

    def rnd(n):
        return aa[rng.integers(0, 20, n)].tobytes().decode()

    def mut(s, d):
        b = np.frombuffer(s.encode(), dtype=np.uint8).copy()
        m = rng.random(len(b)) < d
        b[m] = aa[rng.integers(0, 20, int(m.sum()))]
        return b.tobytes().decode()

    A = rnd(40000)
    recs = [("T0", A), ("T1", mut(A[1000:39000], 0.2)), ("T2", mut(A[20000:33000], 0.1)), ("S0", mut(A[35000:35400], 0.1)),
            ("R", "MKV" * 700)]
    fa = "".join(">%s\n%s\n" % r for r in recs).encode() + synthprot.synthprot(300, 250, 9)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=1000003, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_randomised_differential(fs, oracle, tmp_path, monkeypatch, seed):
    """random workload shapes x flag combinations (the generator of tools/diag/fuzz_parity.py, three draws per seed)"""
    from swiftortho_amd import synthprot
    rng = np.random.default_rng(seed)
    alphas = [oracle.AA9, oracle.AA9 + "/A,KR,EDNQ,C,G,H,ILVM,FYW,P,ST", "A,C,D,E,F,G,H,I,K,L,M,N,P,Q,R,S,T,V,W,Y"]
    seeds = ["111111", "1101011", "111111,1101011", "11111011111", "1110111", "11011"]
    for _ in range(3):
        N, L = int(rng.integers(150, 1200)), int(rng.integers(40, 420))
        gen = synthprot.uniform_proteins if rng.random() < 0.25 else synthprot.synthprot
        fa = gen(N, L, int(rng.integers(1, 1 << 30)))
        kw = dict(ssd=str(rng.choice(seeds)), nr=str(rng.choice(alphas)), ht=int(rng.choice([50021, 1000003, 15000017, 120000000])),
                  chk=int(rng.choice([50000, N // 3 + 1, 97])), step=int(rng.choice([1, 1, 2, 4])), v=int(rng.choice([500, 50, 5, 1200])),
                  expect=float(rng.choice([1e-5, 1e-3, 10.0])), flt=str(rng.choice(["T", "T", "F"])), thr=int(rng.choice([-1, -1, 3, 40])),
                  max_miss=float(rng.choice([1e-3, 0.5])))
        if kw["nr"].count(",") > 15 and kw["ssd"] == "11011":
            kw["ssd"] = "1111111"  # weight-4 seeds on 20 letters: hit lists too long for the CPU oracle
        lo = int(rng.integers(0, N // 2))
        hi = int(min(N, lo + rng.integers(20, 100)))
        monkeypatch.setenv("SOHIT_BATCH", str(int(rng.choice([16384, 37]))))
        monkeypatch.setenv("SOHIT_MAX_HITS", str(int(rng.choice([1 << 30, 50000]))))
        oracle_vs_gpu(fs, oracle, fa, kw, tmp_path, sub=(lo, hi))


BUCKET_CASES = {
    # name: (proteins, length, rng seed, kwargs overrides, SOHIT_BUCKET_AVG)
    "families_narrow_ranges": (2500, 250, 91, dict(), "64"),          # many subject ranges, small buckets
    "families_wide_ranges": (2500, 250, 92, dict(), "100000"),       # one range per chunk: buckets above BG_CAP -> sub-passes
    "multichunk": (1800, 200, 93, dict(chk=700), "256"),
    "uniform": (1500, 300, 94, dict(uniform=True, ht=50021), "512"),  # colliding buckets, no homologs
    "nofilter_thr": (1200, 180, 95, dict(flt="F", thr=7, max_miss=0.5, v=3), "256"),
}


@pytest.mark.parametrize("name", sorted(BUCKET_CASES))
def test_bucketed_binning_forced_vs_oracle(fs, oracle, tmp_path, monkeypatch, name):
    """The sort-free diagonal binning (k_bucket.hip: count / scan / scatter into (query, subject range) buckets, bucket-local LDS
    sort) forced on regardless of the pass-size heuristic, at several range widths; rows, candidates and counters equal the
    oracle's, and the profile shows that the bucketed kernels really ran."""
    from swiftortho_amd import synthprot
    n, ln, seed, over, avg = BUCKET_CASES[name]
    over = dict(over)
    fa = (synthprot.uniform_proteins if over.pop("uniform", False) else synthprot.synthprot)(n, ln, seed)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    kw.update(over)
    monkeypatch.setenv("SOHIT_BUCKET_MIN", "0")
    monkeypatch.setenv("SOHIT_BUCKET_AVG", avg)
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)
    s = fs.Searcher(profile=True, **kw)
    s.load_ref_bytes(fa)
    s.load_queries_bytes(fa)
    s.search().close()
    assert "group.bucket_group" in s.timing() and "group.sort_keys" not in s.timing()
    s.close()
    # the per-bucket best-diagonal reduction (k_rec_count / k_rec_scatter / k_bkt_best) off: bucketed keys, sorted pass records
    monkeypatch.setenv("SOHIT_BUCKET_BEST", "0")
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)
    monkeypatch.delenv("SOHIT_BUCKET_BEST")
    monkeypatch.setenv("SOHIT_BUCKET", "0")   # and the sorted path is still there
    s = fs.Searcher(profile=True, **kw)
    s.load_ref_bytes(fa)
    s.load_queries_bytes(fa)
    s.search().close()
    assert "group.sort_keys" in s.timing() and "group.bucket_group" not in s.timing()
    s.close()


def test_makedb_writes_the_reference_index_files(fs, tmp_path):
    """fsearch.makedb: the chunk indexes built on the GPU, written in the reference's on-disk format (.idx / .soas / .bin,
    fsearch.py:2283-2352), are byte-identical to the files the REAL reference wrote (tests/golden/idx_toy.*: three chunks, two
    seed patterns, colliding buckets)."""
    import shutil
    meta = json.load(open(os.path.join(GOLD, "idx_toy.json")))
    ref = str(tmp_path / "ref.fsa")
    shutil.copyfile(os.path.join(GOLD, "idx_toy.ref.fsa"), ref)
    chunks = fs.makedb(ref, **meta["args"])
    assert len(chunks) == 3
    for suffix in meta["files"]:
        assert open(ref + suffix, "rb").read() == open(os.path.join(GOLD, "idx_toy" + suffix), "rb").read(), suffix


def _bucket_members(s, k, NC):
    start, ent = s.chunk_index(k)
    bucket = np.repeat(np.arange(NC), np.diff(start.astype(np.int64)))
    order = np.lexsort((ent, bucket))
    return start, ent[order]


def test_load_index_reads_the_reference_files(fs, oracle, tmp_path):
    """Fasta.load (fsearch.py:2355-2444): the index files the REAL reference wrote (tests/golden/idx_toy.*: three chunks of 25 sequences,
    two seed patterns, 5003 buckets, so buckets mix patterns and the tag of every slot has to be recovered) are made resident with
    so_load_index and searched: the chunk indexes equal a freshly built one's member for member, and the rows equal the oracle's."""
    import shutil
    meta = json.load(open(os.path.join(GOLD, "idx_toy.json")))
    a = meta["args"]
    ref = str(tmp_path / "ref.fsa")
    shutil.copyfile(os.path.join(GOLD, "idx_toy.ref.fsa"), ref)
    for suffix in meta["files"]:
        shutil.copyfile(os.path.join(GOLD, "idx_toy" + suffix), ref + suffix)
    assert fs.index_params(ref + ".1")["NC"] == a["ht"] and fs.index_params(ref + ".1")["offset"] == 25
    kw = dict(v=500, expect=1e-5, flt="T")
    want_path = str(tmp_path / "o.sc")
    oracle.blastp(ref, ref, want_path, ssd=a["space"], nr=a["nr"], expect=1e-5, v=500, step=a["step"], flt="T", ht=a["ht"], chk=a["chk"])
    want = open(want_path, "rb").read()
    assert want.count(b"\n") > 50
    s = fs.load(ref, **kw)
    s.load_queries(ref)
    hits = s.search()
    got = b"".join(hits.rows())
    hits.close()
    assert s.counters()["n_chunks"] == 3
    fresh = fs.Searcher(ssd=a["space"], nr=a["nr"], ht=a["ht"], chk=a["chk"], step=a["step"], **kw)
    fresh.load_ref(ref)
    fresh.load_queries(ref)
    h2 = fresh.search()
    rows2 = b"".join(h2.rows())
    h2.close()
    for k in range(3):
        assert s.chunk_threshold(k) == fresh.chunk_threshold(k)
        st_l, ent_l = _bucket_members(s, k, a["ht"])
        st_f, ent_f = _bucket_members(fresh, k, a["ht"])
        assert np.array_equal(st_l, st_f) and np.array_equal(ent_l, ent_f), k
        # the file's slot order is kept: descending (subject, tag, pos) inside every bucket
        start, ent = s.chunk_index(k)
        bucket = np.repeat(np.arange(a["ht"]), np.diff(start.astype(np.int64)))
        same = bucket[1:] == bucket[:-1]
        assert np.all(ent[1:][same] < ent[:-1][same])
    assert got == want and rows2 == want
    # a context with other seeds refuses the files; so does another reference
    other = fs.Searcher(ssd="111111", nr=a["nr"], ht=a["ht"], chk=a["chk"], step=1, **kw)
    other.load_ref(ref)
    with pytest.raises(fs.SohitError, match="was built with"):
        other.load_index(ref)
    other.close()
    wrong = str(tmp_path / "wrong.fsa")
    fa = open(ref, "rb").read()
    open(wrong, "wb").write(fa.replace(b"\n", b"\nA", 2))
    with pytest.raises(fs.SohitError, match="does not belong"):
        fs.load(wrong, name=ref, **kw)
    s.close()
    fresh.close()


@pytest.mark.parametrize("space,ht", [("111111", 120000000), ("11111011111,1101011", 1000003)])
def test_makedb_then_load_equals_a_fresh_build(fs, tmp_path, space, ht):
    """makedb -> load round trip on a 3-chunk synthetic set (default bucket count: the 480 MB start table per chunk; and two patterns
    with collisions): rows from the loaded index are byte-identical to those of an index built in memory."""
    from swiftortho_amd import synthprot
    fa = synthprot.synthprot(700, 180, 11)
    ref = str(tmp_path / "r.fsa")
    open(ref, "wb").write(fa)
    chunks = fs.makedb(ref, space=space, nr=fs.AA9, step=1, ht=ht, chk=250)
    assert len(chunks) == 3
    kw = dict(v=500, expect=1e-5, flt="T")
    fresh = fs.Searcher(ssd=space, nr=fs.AA9, ht=ht, chk=250, step=1, **kw)
    fresh.load_ref(ref)
    fresh.load_queries(ref)
    h = fresh.search()
    want = b"".join(h.rows())
    h.close()
    s = fs.load(ref, chk=250, **kw)
    s.load_queries(ref)
    h = s.search()
    got = b"".join(h.rows())
    h.close()
    assert len(want) > 10000 and got == want
    assert [s.chunk_threshold(k) for k in range(3)] == [fresh.chunk_threshold(k) for k in range(3)]
    # drop + rebuild after a load goes back to the built index
    s.drop_index()
    h = s.search()
    assert b"".join(h.rows()) == want
    h.close()
    s.close()
    fresh.close()
    for k in range(3):
        for suf in (".idx", ".soas", ".bin"):
            os.remove("%s.%d%s" % (ref, k, suf))


@pytest.mark.parametrize("name", ["families_narrow_ranges", "multichunk", "uniform"])
@pytest.mark.parametrize("mode", ["2", "0"], ids=["boundaries_checked_against_the_counting_pass", "counting_pass"])
def test_bucket_counts_from_range_boundaries(fs, oracle, tmp_path, monkeypatch, name, mode):
    """The count pass of the bucketed binning reads the range boundaries of the ordered index buckets instead of the entries (order_chunk /
    range_table / k_bkt_count_tab; default).  SOHIT_COUNT_TAB=2 runs the counting pass beside it and fails the search if one cell of the
    (tile, range) matrix differs; 0 is the counting pass alone.  Rows, candidates and counters are the oracle's both ways, and the profile says
    which one ran."""
    from swiftortho_amd import synthprot
    n, ln, seed, over, avg = BUCKET_CASES[name]
    over = dict(over)
    fa = (synthprot.uniform_proteins if over.pop("uniform", False) else synthprot.synthprot)(n, ln, seed)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    kw.update(over)
    monkeypatch.setenv("SOHIT_BUCKET_MIN", "0")
    monkeypatch.setenv("SOHIT_BUCKET_AVG", avg)
    monkeypatch.setenv("SOHIT_COUNT_TAB", mode)
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)
    s = fs.Searcher(profile=True, **kw)
    s.load_ref_bytes(fa)
    s.load_queries_bytes(fa)
    for _ in range(2):   # (the second search finds the chunks ordered and the tables built)
        h = s.search()
        rows = b"".join(h.rows())
        h.close()
        assert rows == open(str(tmp_path / "o.sc"), "rb").read()
    assert ("seed.bucket_count_tab_launches" in s.timing()) == (mode == "2")
    s.close()


@pytest.mark.parametrize("n,ln,avg", [(40, 4300, "100000"), (24, 5200, "64"), (300, 900, "100000")])
def test_bucketed_binning_long_sequences_vs_oracle(fs, oracle, tmp_path, monkeypatch, n, ln, avg):
    """Forced bucketed binning on long proteins: a self hit brings thousands of hits on one diagonal to ONE subject -- segments of
    up to BG_CAP hits ranked by whole waves, and above BG_CAP the kernel refuses and the pass is redone on the sorted path."""
    from swiftortho_amd import synthprot
    fa = synthprot.synthprot(n, ln, 1234 + n)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=1000003, chk=50000, step=1, v=500, expect=1e-5, flt="F")
    monkeypatch.setenv("SOHIT_BUCKET_MIN", "0")
    monkeypatch.setenv("SOHIT_BUCKET_AVG", avg)
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)


def test_exact_threshold_replay(fs, oracle, monkeypatch):
    """the rare exact get_mu_sd replay path gives the same threshold as the integer-sum path"""
    from swiftortho_amd import synthprot
    fa = synthprot.synthprot(300, 150, 83)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=1000003, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    ix = oracle.Index(fa, kw["ssd"], kw["nr"], 1, kw["ht"])
    for forced in (False, True):
        if forced:
            monkeypatch.setenv("SOHIT_EXACT_THRESHOLD", "1")
        s = fs.Searcher(**kw)
        s.load_ref_bytes(fa)
        s.build_index()
        assert s.chunk_threshold(0) == ix.threshold
        s.close()


def test_full_size_properties(fs):
    """BASELINE config 2 (10k x 300 aa self-search, seed 111111): size-independent properties."""
    from swiftortho_amd import synthprot
    fa = synthprot.synthprot(10000, 300)
    kw = dict(ssd="111111", nr="AST,CFILMVY,DN,EQ,G,H,KR,P,W", ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    s, hits, _ = gpu_rows(fs, fa, fa, kw)
    g = hits.array()
    assert len(g) > 10000
    q = g["qidx"]
    assert np.all(np.diff(q) >= 0)                       # queries contiguous and ascending
    same = q[1:] == q[:-1]
    assert np.all(g["bit"][1:][same] <= g["bit"][:-1][same])   # descending bit inside a query
    assert np.all(np.bincount(q) <= 500)                 # at most v rows per query
    assert np.all(g["evalue"] <= 1e-5)
    assert np.all((g["qst"] >= 1) & (g["qed"] <= g["qlen"]) & (g["sst"] >= 1) & (g["sed"] <= g["slen"]))
    assert np.all(g["mis"] + g["matches"] == g["aln"])
    self_hit = g[g["qidx"] == g["sidx"]]
    lens = s.query_lengths()
    assert len(np.unique(self_hit["qidx"])) >= 0.99 * len(lens)   # (masked low-complexity queries may miss themselves)
    # idempotence: a second run over a sub-range reproduces the same records
    h2 = s.search(1234, 1300)
    g2 = h2.array()
    assert g2.tobytes() == g[(g["qidx"] >= 1234) & (g["qidx"] < 1300)].tobytes()
    h2.close()
    hits.close()
    s.close()


RCCL_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from swiftortho_amd import dist as sdist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for n in (0, 80, 80 * 50001):
    recs = (np.arange(n, dtype=np.int64) %% 251).astype(np.uint8)
    g = sdist.gather_device_records(torch.from_numpy(recs).cuda())
    assert g.sizes == [n], g.sizes
    assert g.buf.is_cuda and g.buf.numel() == n
    got = g.arrays()[0]
    assert got.shape == (n,) and np.array_equal(got, recs)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("RCCL_GATHER_OK", float(t.item()))
dist.destroy_process_group()
'''


def test_gather_records_over_rccl(tmp_path):
    """The hit gather of bench.py / find_hit.py -a N on the backend the GPU box really uses ("nccl" == RCCL):
    size all_gather, size-exact device gatherv, pinned device-to-host copy, and the all_reduce / barrier bench.py issues.
    One rank (the box has one GPU); the world_size-2 flow is covered on gloo in test_host_logic.py."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER % root)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200), WORLD_SIZE="1", RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "RCCL_GATHER_OK 1.5" in p.stdout


def test_collapse_search_expand_chain(oracle, tmp_path):
    """scripts/run_all_fast.py's search step (run_all_fast.py:109-118): nr_flt -> bin/find_hit.py on the GPU -> nr2full.  The
    collapsed proteome is the golden made by the REAL nr_flt.py, the search rows equal the oracle's with the flags the wrapper
    passes, and the expanded file equals the (golden-pinned) expansion of the oracle's rows."""
    import shutil
    from swiftortho_amd import nr
    fas = str(tmp_path / "p.fsa")
    shutil.copy(os.path.join(GOLD, "nr_dups.fsa"), fas)
    full = nr.search_collapsed(fas, seed="111111", cpus="1", hits="500")
    assert open(fas + "_nr.fsa").read() == open(os.path.join(GOLD, "nr_dups.nr.fsa")).read()
    want_sc = str(tmp_path / "o.sc")
    oracle.blastp(fas + "_nr.fsa", fas + "_nr.fsa", want_sc, ssd="111111", nr=oracle.AA9, expect=1e-5, v=500, step=1, flt="T", ht=120000000, chk=50000,
                  max_miss=5e-2)
    got_nr = open(os.path.join(fas + "_results", "p.fsa_nr.fsa.sc")).read()
    assert got_nr.count("\n") > 100 and got_nr == open(want_sc).read()
    assert open(full).read() == "".join(l + "\n" for l in nr.nr2full(open(want_sc)))
    assert open(full).read().count("\n") > got_nr.count("\n")


def _check_per_rank(line, world):
    """the multi-rank bench line explains itself: every rank's stage times and the model's terms are there"""
    pr = line["per_rank_ms_per_step"]
    for k in ("index_ms", "search_ms", "gather_ms_incl_wait", "d2h_ms", "query_aa", "queries"):
        assert len(pr[k]) == world, k
    assert sum(pr["query_aa"]) == line["config"]["query_aa"]
    m = pr["model"]
    for k in ("records_bytes", "index_ms_replicated_max", "index_build_decision", "search_ms_max", "search_ms_even_split", "gather_ms_rank0_incl_wait",
              "gather_ms_predicted", "d2h_mode", "d2h_ms_after_gather_max", "d2h_ms_predicted", "step_ms_sum_of_terms"):
        assert k in m and m[k] is not None, k
    assert m["records_bytes"] == 80 * line["config"]["rows"] and m["search_ms_max"] > 0 and "own records" in m["d2h_mode"]


def test_bench_plain_command_two_ranks():
    """The driver's command shape `python bench.py --gpus N ...` with no torch.distributed environment: bench.py starts its
    own ranks.  Two ranks over gloo sharing GPU 0 (the box has one GPU: functional check of the N > 1 flow, not a timing):
    one JSON line with n_gpus 2 and the same row count as the N = 1 line of the same workload."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(SOHIT_BENCH_BACKEND="gloo", SOHIT_BENCH_ONE_GPU="1")
    outs = {}
    for n in (1, 2):
        p = _cli(["bench.py", "--gpus", str(n), "--steps", "1", "--warmup", "0", "--workload", "c2", "--no-cpu-baseline"], env=env)
        assert p.returncode == 0, p.stderr.decode()[-3000:]
        lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout.decode()[-2000:]
        outs[n] = json.loads(lines[0])
        assert outs[n]["n_gpus"] == n and outs[n]["value"] > 0 and outs[n]["scaling"] == "strong"
    _check_per_rank(outs[2], 2)
    assert outs[1]["config"]["rows"] == outs[2]["config"]["rows"] > 10000
    assert outs[1]["config"]["workload"] == outs[2]["config"]["workload"]


@pytest.mark.parametrize("poison", ["0xFF", "0x5A"])
def test_results_do_not_depend_on_stale_device_memory(poison):
    """Every fresh device allocation is filled with a byte pattern (SOHIT_POISON) and a short randomised differential
    runs in that process: regression for the class-array pads k_ungap's windows reach into (a stale 0xFF there once
    read one byte past the LDS score table).  tools/diag/fuzz_parity.py compares rows and candidate lists with the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SOHIT_POISON=poison)
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "diag", "fuzz_parity.py"), "4", "20261002"], env=env, cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    assert p.stdout.count(" ok ") == 4, p.stdout[-3000:]


def _cli(args, **kw):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable] + args, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, **kw)


@pytest.mark.parametrize("name", ["toy_default", "toy_chunks", "toy_multiseed", "example_cfg1"])
def test_fsearch_c_stand_in_cli_matches_golden(tmp_path, name):
    """bin/fsearch-c (the reference's native CLI, fsearch.py:3152-3264) with the golden's own flags: the -o file is the
    golden .sc; split in two -l/-u blocks with -O a (find_hit.py's block scheme, 119-132) it is the same file; with -o
    empty the rows go to stdout (3253)."""
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    ref = os.path.join(GOLD, name + ".ref.fsa")
    qry = os.path.join(GOLD, name + ".qry.fsa") if meta["separate_query"] else ref
    want = open(os.path.join(GOLD, name + ".sc"), "rb").read()
    base = [os.path.join("bin", "fsearch-c"), "-p", "blastp", "-i", qry, "-d", ref] + list(meta["flags"])
    out = tmp_path / "whole.sc"
    p = _cli(base + ["-o", str(out)])
    assert p.returncode == 0, p.stderr[-2000:]
    assert out.read_bytes() == want
    nq = sum(1 for l in open(qry, "rb") if l.startswith(b">"))
    if nq >= 2:
        out2 = tmp_path / "blocks.sc"
        mid = nq // 2
        for lo, hi, mode in ((0, mid, "w"), (mid, nq, "a")):
            p = _cli(base + ["-o", str(out2), "-l", str(lo), "-u", str(hi), "-O", mode])
            assert p.returncode == 0, p.stderr[-2000:]
        assert out2.read_bytes() == want
    p = _cli(base)
    assert p.returncode == 0 and p.stdout == want


@pytest.mark.parametrize("name", ["toy_default", "toy_aa20"])
def test_find_hit_cli_matches_golden(tmp_path, name):
    """bin/find_hit.py with the reference's flag letters (find_hit.py:227-228) writes the golden .sc."""
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    ref = os.path.join(GOLD, name + ".ref.fsa")
    qry = os.path.join(GOLD, name + ".qry.fsa") if meta["separate_query"] else ref
    out = tmp_path / "fh.sc"
    p = _cli([os.path.join("bin", "find_hit.py"), "-p", "blastp", "-i", qry, "-d", ref, "-o", str(out), "-a", "1"] + list(meta["flags"]))
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-2000:])
    assert out.read_bytes() == open(os.path.join(GOLD, name + ".sc"), "rb").read()


ONE_GPU_GLOO = dict(SOHIT_BENCH_BACKEND="gloo", SOHIT_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")  # N ranks share GPU 0


@pytest.mark.parametrize("name", launcher_golden_names())
def test_find_hit_cli_matches_launcher_golden(tmp_path, name):
    """bin/find_hit.py against outputs of the REAL reference launcher (tools/refharness/ref_find_hit.py): more queries
    than references with the default -l/-u (End < 0 -> the QUERY count, find_hit.py:97-105), the -a 3 block scheme whose last
    block runs past -u (107-116; here three ranks over gloo sharing the GPU), and the reference split + `sort -m | awk`
    merge (303-351) with the threshold forced low."""
    import shutil
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    ref = str(tmp_path / "ref.fsa")   # the split path creates <ref>_parts next to the file
    shutil.copyfile(os.path.join(GOLD, name + ".ref.fsa"), ref)
    qry = os.path.join(GOLD, name + ".qry.fsa")
    out = tmp_path / "fh.sc"
    env = dict(os.environ, **ONE_GPU_GLOO)
    if meta["max_chr"] is not None:
        env["SWIFTORTHO_MAX_CHR"] = str(meta["max_chr"])
    p = _cli([os.path.join("bin", "find_hit.py"), "-p", "blastp", "-i", qry, "-d", ref, "-o", str(out)] + list(meta["find_hit_flags"]), env=env)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    assert out.read_bytes() == open(os.path.join(GOLD, name + ".sc"), "rb").read()
    assert not os.path.exists(ref + "_parts")


def test_two_rank_search_equals_one_rank(tmp_path):
    """find_hit.py -a 2 (two ranks, query shards, hit gather; gloo with both ranks on GPU 0) writes the file of -a 1."""
    from swiftortho_amd import synthprot
    fa = tmp_path / "w.fsa"
    fa.write_bytes(synthprot.synthprot(3000, 300, 77))
    outs = []
    for a in (1, 2):
        out = tmp_path / ("a%d.sc" % a)
        p = _cli([os.path.join("bin", "find_hit.py"), "-p", "blastp", "-i", str(fa), "-d", str(fa), "-o", str(out), "-e", "1e-5", "-s", "111111",
                  "-a", str(a)], env=dict(os.environ, **ONE_GPU_GLOO))
        assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
        outs.append(out.read_bytes())
    assert outs[0].count(b"\n") > 5000
    assert outs[0] == outs[1]


def test_two_rank_search_over_rccl_on_two_gpus(tmp_path):
    """The real multi-GPU flow -- one rank per GPU, hit records gathered device to device over RCCL -- whenever the box has two GPUs
    (the pool's boxes have one: skipped there; the gloo / one-GPU test above covers the same code path functionally)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from swiftortho_amd import synthprot
    fa = tmp_path / "w.fsa"
    fa.write_bytes(synthprot.synthprot(3000, 300, 77))
    outs = []
    env = {k: v for k, v in os.environ.items() if k not in ("SOHIT_BENCH_BACKEND", "SOHIT_BENCH_ONE_GPU")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    for a in (1, 2):
        out = tmp_path / ("a%d.sc" % a)
        p = _cli([os.path.join("bin", "find_hit.py"), "-p", "blastp", "-i", str(fa), "-d", str(fa), "-o", str(out), "-e", "1e-5", "-s", "111111",
                  "-a", str(a)], env=env)
        assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
        outs.append(out.read_bytes())
    assert outs[0].count(b"\n") > 5000 and outs[0] == outs[1]
    # ... and the bench's strong-scaling flow over RCCL: one JSON line that carries every rank's terms
    p = _cli(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "c2", "--no-cpu-baseline"], env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0
    _check_per_rank(line, 2)


@pytest.mark.parametrize("spec", ["0", "1"])
def test_device_resident_results_and_query_work(fs, monkeypatch, spec):
    """so_search_device leaves the same 80-byte so_hit records in HBM that so_search_loaded returns on the host (identity and
    e-value evaluated on the device: bit-equal doubles); so_query_work's per-query counts add up to the seed hits searched.
    Both with and without the speculative traces of the first aligner round."""
    import torch
    from swiftortho_amd import synthprot
    monkeypatch.setenv("SOHIT_SPEC", spec)
    fa = synthprot.synthprot(1500, 200, 3)
    kw = dict(ssd="111111", nr="AST,CFILMVY,DN,EQ,G,H,KR,P,W", ht=120000000, chk=700, step=1, v=500, expect=1e-5, flt="T")
    s = fs.Searcher(**kw)
    s.load_ref_bytes(fa)
    s.load_queries_bytes(fa)
    for lo, hi in ((-1, -1), (100, 433), (7, 7)):
        h = s.search(lo, hi)
        host = h.raw_bytes()
        h.close()
        d = s.search_device(lo, hi)
        assert len(d) * d.record_bytes == len(host)
        assert d.tensor().cpu().numpy().tobytes() == host
    w = s.query_work()
    assert len(w) == 1500 and w.min() >= 0
    s.reset_counters()
    s.search().close()
    assert int(w.sum()) == s.counters()["seed_hits"]
    assert int(s.query_work(100, 433).sum()) == int(w[100:433].sum())
    s.close()


def _properties(g, v=500, expect=1e-5):
    q = g["qidx"]
    assert np.all(np.diff(q) >= 0)
    same = q[1:] == q[:-1]
    assert np.all(g["bit"][1:][same] <= g["bit"][:-1][same])
    assert np.all(np.bincount(q - q.min()) <= v)
    assert np.all(g["evalue"] <= expect)
    assert np.all((g["qst"] >= 1) & (g["qed"] <= g["qlen"]) & (g["sst"] >= 1) & (g["sed"] <= g["slen"]))
    assert np.all(g["mis"] + g["matches"] == g["aln"])


def _sampled_oracle(s, oracle, fa_path, kw, ranges):
    """per-query results do not depend on the query partition, so an oracle run of -l/-u sub-ranges against the whole
    reference checks a sample of a search that is too large to replay on the CPU"""
    for lo, hi in ranges:
        out = fa_path + ".%d.sc" % lo
        oracle.blastp(fa_path, fa_path, out, ssd=kw["ssd"], nr=kw["nr"], expect=kw["expect"], v=kw["v"], step=kw["step"], flt=kw["flt"],
                      ht=kw["ht"], chk=kw["chk"], st=lo, ed=hi)
        h = s.search(lo, hi)
        rows = b"".join(h.rows())
        h.close()
        want = open(out, "rb").read()
        assert want.count(b"\n") > 0
        assert rows == want, "queries [%d, %d) differ from the oracle" % (lo, hi)


def test_config3_full_size(fs, oracle, tmp_path):
    """BASELINE config 3: 100k proteins x 300 aa self-search, seed 11111011111, two reference chunks of 50k.
    Full run: size-independent properties; two sampled query ranges (one per chunk's taxa): identical to the oracle."""
    from swiftortho_amd import synthprot
    fa = synthprot.synthprot(100000, 300)
    kw = dict(ssd="11111011111", nr=oracle.AA9, ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    s, hits, _ = gpu_rows(fs, fa, fa, kw)
    assert s.counters()["n_chunks"] == 2
    g = hits.array()
    assert len(g) > 1000000
    _properties(g)
    self_hit = g[g["qidx"] == g["sidx"]]
    assert len(np.unique(self_hit["qidx"])) >= 0.99 * 100000
    h2 = s.search(61234, 61300)   # idempotence on a sub-range
    assert h2.array().tobytes() == g[(g["qidx"] >= 61234) & (g["qidx"] < 61300)].tobytes()
    h2.close()
    hits.close()
    p = str(tmp_path / "c3.fsa")
    open(p, "wb").write(fa)
    _sampled_oracle(s, oracle, p, kw, [(777, 793), (88000, 88016)])
    s.close()


def test_config4_shape_sampled(fs, oracle, tmp_path):
    """BASELINE config 4's shape at a size the suite affords: 200k proteins, four resident 50k chunks, seed 111111
    (1.6e5 seed hits per query and chunk).  16 sampled queries against all four chunks: identical to the oracle; a
    512-query range: properties."""
    from swiftortho_amd import synthprot
    fa = synthprot.synthprot(200000, 300)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    s = fs.Searcher(**kw)
    s.load_ref_bytes(fa)
    s.load_queries_bytes(fa)
    h = s.search(150000, 150512)
    assert s.counters()["n_chunks"] == 4
    g = h.array()
    assert len(g) > 5000
    _properties(g)
    h.close()
    p = str(tmp_path / "c4.fsa")
    open(p, "wb").write(fa)
    _sampled_oracle(s, oracle, p, kw, [(123450, 123458), (199990, 199998)])
    s.close()


def test_config4_full_size(fs, oracle, tmp_path):
    """BASELINE config 4 at its own size: 1 M proteins x 300 aa (300 M aa), seed 111111, twenty resident 50k chunks.
    A 512-query range against all 1 M references: size-independent properties, idempotence, sub-range invariance; two
    8-query ranges (first and last taxa of the set) against `oracle/sohit_cpu -l/-u` over the whole reference: identical
    bytes.  (The whole 1 M x 1 M job is 2e12 seed hits -- two minutes of GPU, weeks of CPU: tools/diag/run_config.py.)"""
    from swiftortho_amd import synthprot
    fa = synthprot.synthprot(1000000, 300)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    s = fs.Searcher(**kw)
    s.load_ref_bytes(fa)
    s.load_queries_bytes(fa)
    s.reset_counters()
    h = s.search(500000, 500512)
    c = s.counters()
    assert c["n_chunks"] == 20 and c["ref_seqs"] == 1000000
    assert c["seed_hits"] > 512 * 20 * 50000   # > 5e4 index entries visited per query and chunk
    g = h.array()
    assert len(g) > 50000
    _properties(g)
    assert len(np.unique(g["qidx"][g["qidx"] == g["sidx"]])) >= 0.99 * 512
    h2 = s.search(500100, 500164)   # per-query results do not depend on the query partition
    assert h2.array().tobytes() == g[(g["qidx"] >= 500100) & (g["qidx"] < 500164)].tobytes()
    h2.close()
    h.close()
    p = str(tmp_path / "c4.fsa")
    open(p, "wb").write(fa)
    del fa
    _sampled_oracle(s, oracle, p, kw, [(1234, 1242), (999990, 999998)])
    s.close()


def _ragged_queries(base):
    recs, cur = {}, None
    for l in base.decode().strip().split("\n"):
        if l.startswith(">"):
            cur = l
            recs[cur] = ""
        else:
            recs[cur] += l
    names = list(recs)
    s0, s1 = recs[names[0]], recs[names[1]]
    return b"".join([
        (names[0] + " some description here\n").encode(), (s0[:50] + "\n" + s0[50:] + "\n").encode(),  # wrapped lines, description
        b">tiny\nMKV\n",                                                  # shorter than every seed: no hits (our definition)
        b">empty\n",                                                      # empty record: skipped
        b">allx\n" + b"X" * 40 + b"\n",                                   # every window rejected
        b">lower\n" + s1.lower().encode() + b"\n",                        # lower case scores like upper case, hashes differently
        b">crlf\r\n" + s1[:60].encode() + b"\r\n" + s1[60:].encode() + b"\r\n",  # '\r' is kept as a residue / in the id (F1)
        b">weird\n" + (s0[:30] + "*U-J" + s0[30:]).encode() + b"\n",      # bytes outside the 23 letters score -4
        b">last_no_newline\n" + s0.encode(),                              # file ends without '\n'
    ])


@pytest.mark.parametrize("kw", [dict(ssd="111111", expect=1e-3, v=500, step=1, ht=1000003, chk=50000),
                                dict(ssd="1101011", expect=10.0, v=5, step=2, ht=50021, chk=17, flt="F")])
def test_ragged_inputs_vs_oracle(fs, oracle, tmp_path, kw):
    """Empty, tiny, all-X, lower-case, CRLF, non-alphabet and unterminated query records against a small family set."""
    from swiftortho_amd import synthprot
    base = synthprot.synthprot(60, 120, 5)
    qry = _ragged_queries(base)
    qp, rp, op = tmp_path / "q.fsa", tmp_path / "r.fsa", tmp_path / "o.sc"
    qp.write_bytes(qry), rp.write_bytes(base)
    r = oracle.blastp(str(qp), str(rp), str(op), **kw)
    want = op.read_bytes()
    assert len(want) > 0 and r.nqueries == 8
    s, hits, rows = gpu_rows(fs, base, qry, dict(nr=oracle.AA9, thr=-1, max_miss=1e-3, flt="T", **kw) if "flt" not in kw else
                             dict(nr=oracle.AA9, thr=-1, max_miss=1e-3, **kw))
    if rows != want:
        a, b = rows.split(b"\n"), want.split(b"\n")
        for i in range(max(len(a), len(b))):
            x = a[i] if i < len(a) else b"<none>"
            y = b[i] if i < len(b) else b"<none>"
            assert x == y, "row %d differs\n gpu: %r\n ref: %r" % (i, x, y)
    hits.close()
    s.close()


def test_no_hits_and_empty_sets(fs, tmp_path):
    """Unrelated query, queries that cannot seed, empty query file, empty reference: no rows, no crash."""
    from swiftortho_amd import synthprot
    ref = synthprot.uniform_proteins(50, 80, 11)
    kw = dict(ssd="11111011111", nr="AST,CFILMVY,DN,EQ,G,H,KR,P,W", ht=1000003, chk=50000, step=1, v=500, thr=-1, expect=1e-30, max_miss=1e-3, flt="T")
    for qry in (synthprot.uniform_proteins(5, 70, 12), b">tiny\nMK\n>empty\n", b""):
        s, hits, rows = gpu_rows(fs, ref, qry, kw)
        assert rows == b"" and len(hits) == 0
        hits.close()
        s.close()
    s, hits, rows = gpu_rows(fs, b"", ref, kw)  # empty reference: nothing to hit
    assert rows == b"" and len(hits) == 0
    hits.close()
    s.close()


def test_fasta_quirks_vs_oracle(fs, oracle, tmp_path):
    """Record parsing exactly as fsearch.py:1543-1553 / 2182-2205: '>' only starts a record at a line start, blank lines and
    blanks inside sequence lines stay residues or vanish as the reference's join makes them, ids end at the first space
    (tabs stay), duplicate ids and a 300-character id, and the same records on the reference side (column 16 = whole header)."""
    from swiftortho_amd import synthprot
    base = synthprot.synthprot(60, 120, 5)
    recs, cur = {}, None
    for l in base.decode().strip().split("\n"):
        if l.startswith(">"):
            cur = l
            recs[cur] = ""
        else:
            recs[cur] += l
    names = list(recs)
    s0, s1, s2 = recs[names[0]], recs[names[1]], recs[names[2]]
    rag = b"".join([
        b">gt_inside\n" + (s0[:40] + ">" + s0[40:]).encode() + b"\n",
        b">blank_lines\n" + s1[:30].encode() + b"\n\n" + s1[30:].encode() + b"\n\n",
        b">spaces in\theader\twith tabs\n" + (s2[:20] + "  " + s2[20:70] + " " + s2[70:]).encode() + b"\n",
        b">dup\n" + s0.encode() + b"\n", b">dup\n" + s0.encode() + b"\n",
        b">" + b"L" * 300 + b" longheader\n" + s1.encode() + b"\n",
    ])
    ref = base + rag
    qp, rp, op = tmp_path / "q.fsa", tmp_path / "r.fsa", tmp_path / "o.sc"
    qp.write_bytes(rag), rp.write_bytes(ref)
    kw = dict(ssd="111111", expect=1e-3, v=500, step=1, ht=1000003, chk=50000)
    r = oracle.blastp(str(qp), str(rp), str(op), **kw)
    want = op.read_bytes()
    assert r.nqueries == 6 and want.count(b"\n") > 12
    s, hits, rows = gpu_rows(fs, ref, rag, dict(nr=oracle.AA9, thr=-1, max_miss=1e-3, flt="T", **kw))
    if rows != want:
        a, b = rows.split(b"\n"), want.split(b"\n")
        for i in range(max(len(a), len(b))):
            x = a[i] if i < len(a) else b"<none>"
            y = b[i] if i < len(b) else b"<none>"
            assert x == y, "row %d differs\n gpu: %r\n ref: %r" % (i, x, y)
    hits.close()
    s.close()


FLAG_EXTREMES = [
    dict(rst=10, red=40), dict(rst=55, red=1000), dict(rst=0, red=1),
    dict(chk=1), dict(chk=2), dict(chk=7, rst=3, red=33),
    dict(v=1), dict(v=2, expect=10.0), dict(v=1000),
    dict(step=50), dict(step=500), dict(step=3, ssd="1101011,111111"),
    dict(st=-1, ed=5), dict(st=55, ed=1000), dict(st=70, ed=80), dict(st=30, ed=10), dict(st=59, ed=60),
    dict(expect=0.0), dict(expect=1e300), dict(expect=1e-300),
    dict(max_miss=0.0), dict(max_miss=5.0), dict(max_miss=0.9999),
    dict(thr=0), dict(thr=1), dict(thr=100000),
    dict(ht=2), dict(ht=7), dict(ht=257),
    dict(flt="F", v=3, chk=11, step=2),
]


@pytest.mark.parametrize("over", FLAG_EXTREMES, ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
def test_flag_extremes_vs_oracle(fs, oracle, tmp_path, over):
    """Corner values of every flag of the path (-L/-U reference ranges, one-sequence chunks, -v 1, steps longer than the
    sequences, query ranges past the end or reversed, e-value bounds 0 / 1e300, -m outside (0, 1), -t 0 / huge, 2-bucket
    tables): rows identical to the oracle's."""
    from swiftortho_amd import synthprot
    fa = synthprot.synthprot(60, 110, 9)
    p = tmp_path / "x.fsa"
    p.write_bytes(fa)
    o = dict(ssd="111111", nr=oracle.AA9, expect=1e-3, v=500, max_miss=1e-3, st=-1, ed=-1, rst=-1, red=-1, thr=-1, step=1, flt="T", ht=1000003,
             chk=50000)
    o.update(over)
    out = tmp_path / "o.sc"
    oracle.blastp(str(p), str(p), str(out), **o)
    want = out.read_bytes()
    s = fs.Searcher(ssd=o["ssd"], nr=o["nr"], ht=o["ht"], chk=o["chk"], step=o["step"], v=o["v"], thr=o["thr"], expect=o["expect"],
                    max_miss=o["max_miss"], flt=o["flt"])
    s.load_ref_bytes(fa, o["rst"], o["red"])
    s.load_queries_bytes(fa)
    hits = s.search(o["st"], o["ed"])
    rows = b"".join(hits.rows())
    if rows != want:
        a, b = rows.split(b"\n"), want.split(b"\n")
        assert len(a) == len(b), "gpu %d rows, oracle %d rows" % (len(a) - 1, len(b) - 1)
        for i in range(len(a)):
            assert a[i] == b[i], "row %d differs\n gpu: %r\n ref: %r" % (i, a[i], b[i])
    hits.close()
    s.close()


def test_tile_boundary_lengths_vs_oracle(fs, oracle, tmp_path):
    """Sequence lengths on both sides of the 4096-residue tile size of kswat_st_long (fsearch.py:1480-1498, 3068): 4095 /
    4096 / 4097 / 8191 / 8192 / 8193, each with a mutated copy, a prefix and a suffix, all against all."""
    from swiftortho_amd import synthprot
    rng = np.random.default_rng(17)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)

    def rnd(n):
        return aa[rng.integers(0, 20, n)].tobytes().decode()

    def mut(s, d):
        b = np.frombuffer(s.encode(), dtype=np.uint8).copy()
        m = rng.random(len(b)) < d
        b[m] = aa[rng.integers(0, 20, int(m.sum()))]
        return b.tobytes().decode()

    recs = []
    for n in (4095, 4096, 4097, 8191, 8192, 8193):
        a = rnd(n)
        recs += [("a%d" % n, a), ("m%d" % n, mut(a, 0.12)), ("p%d" % n, mut(a[:n - 4096 + 300], 0.05) if n > 4096 else mut(a[:3000], 0.05)),
                 ("s%d" % n, mut(a[4090:], 0.05) if n > 4400 else mut(a[3700:], 0.05))]
    fa = "".join(">%s\n%s\n" % r for r in recs).encode() + synthprot.synthprot(40, 200, 3)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=1000003, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)


@pytest.mark.parametrize("flt", ["F", "T"])
def test_tandem_repeats_and_homopolymers_vs_oracle(fs, oracle, tmp_path, flt):
    """Homopolymers and tandem repeats of period 2..11 (unmasked with -F F): hundreds of seed hits per diagonal, every
    diagonal of a pair populated, long chains of overlapping segments, duplicate (qst, sst) pairs across seed patterns."""
    from swiftortho_amd import synthprot
    rng = np.random.default_rng(23)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    recs = []
    for k, period in enumerate((1, 1, 2, 3, 5, 7, 11, 4, 9)):
        unit = "".join(aa[int(x)] for x in rng.integers(0, 20, period))
        for c in range(3):
            n = int(rng.integers(120, 420))
            s = (unit * (n // period + 1))[:n]
            b = list(s)
            for p in rng.integers(0, n, size=n // 25):  # a few point changes so copies differ
                b[int(p)] = aa[int(rng.integers(0, 20))]
            recs.append(("rep%d_%d_%d" % (period, k, c), "".join(b)))
    fa = "".join(">%s\n%s\n" % r for r in recs).encode() + synthprot.synthprot(60, 150, 8)
    kw = dict(ssd="111111,1101011", nr=oracle.AA9, ht=1000003, chk=20, step=1, v=500, expect=1e-3, flt=flt)
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)


@pytest.mark.parametrize("v", [500, 50])
def test_massive_score_ties_vs_oracle(fs, oracle, tmp_path, v):
    """700 identical copies of one protein plus near-copies: every candidate of a query has the same ungapped score and
    most alignments the same bit score, so the reported set and its order are decided purely by the reference's non-stable
    quicksort (fsearch.py:260-327) on ties -- beyond the vmax = max(100, v + 100, 1.1 v) cut as well."""
    from swiftortho_amd import synthprot
    rng = np.random.default_rng(31)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    core = "".join(aa[int(x)] for x in rng.integers(0, 20, 180))
    recs = [("same%04d" % i, core) for i in range(700)]
    for i in range(60):
        b = list(core)
        for p in rng.integers(0, len(b), size=int(rng.integers(1, 4))):
            b[int(p)] = aa[int(rng.integers(0, 20))]
        recs.append(("near%03d" % i, "".join(b)))
    order = rng.permutation(len(recs))
    fa = "".join(">%s\n%s\n" % recs[int(i)] for i in order).encode() + synthprot.synthprot(40, 150, 2)
    kw = dict(ssd="111111", nr=oracle.AA9, ht=1000003, chk=300, step=1, v=v, expect=1e-5, flt="T")
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path, sub=(0, 120))


@pytest.mark.parametrize("lens", [(6,), (7,), (6, 6), (30,), (12, 6, 40), (6, 7, 8, 9, 10, 11, 12)])
def test_micro_references_vs_oracle(fs, oracle, tmp_path, lens):
    """References holding one to a handful of seed windows (the last index slot is never read, fsearch.py:2277 / 2539;
    offset-0 seeds resolve to the previous sequence, 2638-2642): same rows as the oracle, usually none."""
    rng = np.random.default_rng(sum(lens) * 7 + len(lens))
    aa = "ACDEFGHIKLMNPQRSTVWY"
    core = "".join(aa[int(x)] for x in rng.integers(0, 20, 64))
    recs = [("s%d" % i, core[:n]) for i, n in enumerate(lens)]
    fa = "".join(">%s\n%s\n" % r for r in recs).encode()
    for kw in (dict(ssd="111111", nr=oracle.AA9, ht=1000003, chk=50000, step=1, v=500, expect=10.0, flt="F"),
               dict(ssd="111111", nr=oracle.AA9, ht=13, chk=1, step=1, v=500, expect=1e300, flt="F", thr=1)):
        oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)

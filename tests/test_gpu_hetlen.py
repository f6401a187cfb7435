"""GPU parity on LENGTH-HETEROGENEOUS protein sets (round 4): the packed 16-bit aligner is chosen per alignment task, the
bucketed diagonal binning per seed pass of one query length class over diagonal BANDS (k_encode_band32) -- one sequence far above
the fixed widths no longer switches a whole batch to the slow kernels.  Every case here holds sequences above 740 residues (packed
aligner range), above 2048 (hit-word range of the pre-round-4 binning) and above 4096 (the aligner's tiled path) in ONE batch, is
compared with the oracle row for row and candidate for candidate (the 30 000-residue query's k-mer order comes from the side stream in
the default setting), and checks with the library's counters that both variants of
both stages actually ran.

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import numpy as np
import pytest

import json
import os

from conftest import GOLD, het_golden_names
from test_gpu_parity import flags_to_kwargs, fs, oracle_vs_gpu, gpu_rows  # noqa: F401  (fs is a fixture)

pytestmark = pytest.mark.gpu

AA9 = "AST,CFILMVY,DN,EQ,G,H,KR,P,W"


class _CachedOracle:
    """the oracle's end-to-end run of one (input, flags) pair is shared by the path-switch variants of a test (40 s of CPU each)"""

    def __init__(self, real):
        self._real, self._cache = real, {}

    def __getattr__(self, k):
        return getattr(self._real, k)

    def blastp(self, qry, ref, out_path, **kw):
        import hashlib
        key = (hashlib.md5(open(qry, "rb").read()).hexdigest(), tuple(sorted(kw.items())))
        if key not in self._cache:
            r = self._real.blastp(qry, ref, out_path, **kw)
            self._cache[key] = (r, open(out_path, "rb").read())
        r, text = self._cache[key]
        open(out_path, "wb").write(text)
        return r


@pytest.fixture(scope="module")
def coracle(oracle):
    return _CachedOracle(oracle)


def het_fasta(n, seed):
    from swiftortho_amd import synthprot
    return synthprot.synthprot(n, seed=seed, lengths="lognormal")


def lengths_of(fa):
    return np.array([len(x) for x in fa.split(b"\n")[1::2]])


@pytest.mark.parametrize("name", het_golden_names())
@pytest.mark.parametrize("whole", [False, True], ids=["by_range", "one_search"])
def test_heterogeneous_lengths_golden_from_the_real_reference(fs, name, whole):
    """tests/golden/het_*: 300 proteins of log-normal length, subjects of 4562, 4597 and 30 014 residues, searched by the REAL
    reference for every query below 4096 residues (-l/-u ranges; longer queries crash it at fsearch.py:1396).  by_range: the HIP
    path searches the same ranges; one_search: it searches ALL queries in one batch (the giants among them, in their own length
    class) and the rows of the reference-defined queries are cut out of that result -- a query's rows do not depend on its batch."""
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    ref = open(os.path.join(GOLD, name + ".ref.fsa"), "rb").read()
    want = open(os.path.join(GOLD, name + ".sc"), "rb").read()
    kw = flags_to_kwargs(meta["flags"])
    if whole:
        s, hits, rows = gpu_rows(fs, ref, ref, kw)
        skipped = set(meta["skipped_queries"])
        got = b"".join(l + b"\n" for l in rows.split(b"\n")[:-1] if int(l.split(b"\t")[14]) not in skipped)
        assert sum(1 for l in rows.split(b"\n")[:-1] if int(l.split(b"\t")[14]) in skipped) > 0   # the giants do report rows of their own
        hits.close(), s.close()
    else:
        got = b""
        for lo, hi in meta["ranges"]:
            s, hits, rows = gpu_rows(fs, ref, ref, kw, st=lo, ed=hi)
            got += rows
            hits.close(), s.close()
    if got != want:
        a, b = got.split(b"\n"), want.split(b"\n")
        for i in range(max(len(a), len(b))):
            x = a[i] if i < len(a) else b"<none>"
            y = b[i] if i < len(b) else b"<none>"
            assert x == y, "row %d differs\n gpu: %s\n ref: %s" % (i, x.decode("latin-1"), y.decode("latin-1"))


def test_generator_shape():
    fa = het_fasta(3000, 11)
    ln = lengths_of(fa)
    assert len(ln) == 3000 and ln.max() > 29000 and 200 < np.median(ln) < 350
    assert (ln > 740).sum() > 50 and (ln > 2048).sum() >= 3


@pytest.mark.parametrize("env", [{}, {"SOHIT_BUCKET_MIN": "0"}, {"SOHIT_BANDS": "0"}, {"SOHIT_QCLASS": "0"}, {"SOHIT_ALIGN_PK": "0"},
                                 {"SOHIT_BUCKET_MIN": "0", "SOHIT_BUCKET_BEST": "0"}, {"SOHIT_BUCKET_MIN": "0", "SOHIT_POISON": "0xFF"},
                                 {"SOHIT_BATCH": "700", "SOHIT_BUCKET_MIN": "0", "SOHIT_MAX_HITS": "300000"},
                                 {"SOHIT_TRACE_WAVE_ROWS": "16", "SOHIT_TRACE_WAVE_MAX": "100000000", "SOHIT_POISON": "0x3C"},
                                 {"SOHIT_TRACE_WAVE_ROWS": "0"}],
                         ids=["default", "bucket_forced", "no_bands", "no_classes", "no_packed", "bucket_sortbest", "poison", "small_batches",
                              "every_walk_by_a_wave", "every_walk_by_a_thread"])
def test_mixed_lengths_vs_oracle(fs, coracle, tmp_path, monkeypatch, env):
    """3000 proteins, median 277 residues, a tail to 5000 and one of 30 000: rows, candidate lists and counters equal the
    oracle's, whatever the path switches say."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    fa = het_fasta(3000, 11)
    kw = dict(ssd="111111", nr=AA9, ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    c, _ = oracle_vs_gpu(fs, coracle, fa, kw, tmp_path)
    assert c["rows"] > 3000
    if env.get("SOHIT_BUCKET_MIN") == "0" and "SOHIT_BANDS" not in env:
        # queries below 2048 residues (two length classes) are binned by the bucketed passes, the longer ones by the sorted path
        assert 0.5 * c["seed_hits"] < c["hits_bucketed"] < c["seed_hits"]
    if "SOHIT_ALIGN_PK" not in env:
        assert 0 < c["align_wide"] < 0.5 * c["alignments"] and 0 < c["cells_wide"] < c["cells"]


def test_sparse_seed_searches_neighbouring_length_classes_in_one_pass(fs, coracle, tmp_path, monkeypatch):
    """With a long seed the queries visit few index entries each: the length classes would all take the sorted path, so seed_stage
    searches them as ONE pass (the class still waiting for its k-mer orders apart).  Same rows and candidates as the oracle with and
    without the merge, and fewer seed passes with it."""
    fa = het_fasta(3000, 11)
    kw = dict(ssd="11111011111", nr=AA9, ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    launches = {}
    for merge in ("1", "0"):
        monkeypatch.setenv("SOHIT_PASS_MERGE", merge)
        c, _ = oracle_vs_gpu(fs, coracle, fa, kw, tmp_path)
        assert c["rows"] > 3000 and c["hits_bucketed"] == 0
        launches[merge] = c["seed_passes"]
    assert launches["1"] < launches["0"]


def test_mixed_lengths_multi_chunk_two_seeds(fs, oracle, tmp_path, monkeypatch):
    """several chunks (each with its own band numbering) and two seed patterns (no multi-band subjects: first-touch keys walk the hits)"""
    monkeypatch.setenv("SOHIT_BUCKET_MIN", "0")
    fa = het_fasta(1200, 5)
    kw = dict(ssd="111111,1101011", nr=AA9, ht=1000003, chk=500, step=1, v=500, expect=1e-5, flt="T")
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)
    kw = dict(ssd="111111", nr=AA9, ht=1000003, chk=500, step=1, v=500, expect=1e-5, flt="F")
    oracle_vs_gpu(fs, oracle, fa, kw, tmp_path)


def test_device_records_and_query_work_in_file_order(fs, monkeypatch):
    """The batch holds its queries in length-class order; host rows, device-resident records and so_query_work come back in file order."""
    monkeypatch.setenv("SOHIT_BUCKET_MIN", "0")
    fa = het_fasta(1500, 23)
    kw = dict(ssd="111111", nr=AA9, ht=120000000, chk=600, step=1, v=500, expect=1e-5, flt="T")
    s = fs.Searcher(**kw)
    s.load_ref_bytes(fa)
    s.load_queries_bytes(fa)
    for lo, hi in ((-1, -1), (100, 933)):
        h = s.search(lo, hi)
        g = h.array()
        assert np.all(np.diff(g["qidx"]) >= 0) and len(g) > 500
        host = h.raw_bytes()
        h.close()
        d = s.search_device(lo, hi)
        assert d.tensor().cpu().numpy().tobytes() == host
    w = s.query_work()
    ln = lengths_of(fa)
    assert len(w) == 1500 and np.corrcoef(w, ln)[0, 1] > 0.5   # work follows the query's length: file order kept
    s.reset_counters()
    s.search().close()
    assert int(w.sum()) == s.counters()["seed_hits"]
    # the same rows whether or not the queries are reordered inside the batch
    a = s.search()
    rows = b"".join(a.rows())
    a.close()
    s.close()
    monkeypatch.setenv("SOHIT_QCLASS", "0")
    s2, h2, rows2 = gpu_rows(fs, fa, fa, kw)
    assert rows2 == rows
    h2.close()
    s2.close()


def test_out_of_memory_in_phase2_reruns_the_batch_in_halves(fs):
    """A device allocation failing late in a batch (phase 2, after the traced alignments: SOHIT_TEST_OOM_PHASE2 makes the first
    multi-query batch of a process fail there) sends the batch through search_loaded()'s halving path: same rows, same device
    records, counters not double counted.  Run in a child process (the hook fires once per process)."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
from swiftortho_amd import fsearch, synthprot
fa = synthprot.synthprot(900, seed=3, lengths="lognormal")
kw = dict(ssd="111111", nr="AST,CFILMVY,DN,EQ,G,H,KR,P,W", ht=120000000, chk=400, step=1, v=500, expect=1e-5, flt="T")
def run(device):
    s = fsearch.Searcher(**kw)
    s.load_ref_bytes(fa); s.load_queries_bytes(fa)
    if device:
        d = s.search_device(); raw = d.tensor().cpu().numpy().tobytes()
    else:
        h = s.search(); raw = h.raw_bytes(); h.close()
    c = s.counters(); s.close()
    return raw, c
want, c0 = run(False)
os.environ["SOHIT_TEST_OOM_PHASE2"] = "1"
got, c1 = run(int(sys.argv[1]))
assert got == want and len(got) > 80 * 500, (len(got), len(want))
for k in ("rows", "seed_hits", "candidates", "alignments", "n_queries"):
    assert c0[k] == c1[k], (k, c0[k], c1[k])
print("OOM_RERUN_OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for device in ("0", "1"):
        p = subprocess.run([sys.executable, "-c", code, device], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "OOM_RERUN_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])

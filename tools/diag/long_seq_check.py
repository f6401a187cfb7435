import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["SOHIT_KEEP_CANDS"] = "1"
from swiftortho_amd import fsearch, synthprot
from oracle import oracle
oracle.build()
rng = np.random.default_rng(3)
aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
def rnd(n): return aa[rng.integers(0, 20, n)].tobytes().decode()
def mut(s, d):
    b = np.frombuffer(s.encode(), dtype=np.uint8).copy(); m = rng.random(len(b)) < d
    b[m] = aa[rng.integers(0, 20, int(m.sum()))]; return b.tobytes().decode()
A = rnd(40000)
recs = [("T0", A), ("T1", mut(A[1000:39000], 0.2)), ("T2", mut(A[20000:33000], 0.1)), ("S0", mut(A[35000:35400], 0.1)), ("R", ("MKV" * 700))]
fa = "".join(">%s\n%s\n" % r for r in recs).encode() + synthprot.synthprot(300, 250, 9)
d = tempfile.mkdtemp(); p = os.path.join(d, "x.fsa"); open(p, "wb").write(fa)
kw = dict(ssd="111111", nr=oracle.AA9, ht=1000003, chk=50000, step=1, v=500, expect=1e-5, flt="T")
r = oracle.blastp(p, p, os.path.join(d, "o.sc"), ssd=kw["ssd"], nr=kw["nr"], expect=kw["expect"], v=kw["v"], step=kw["step"], flt=kw["flt"], ht=kw["ht"], chk=kw["chk"], st=-1, ed=-1)
s = fsearch.Searcher(**kw); s.load_ref_bytes(fa); s.load_queries_bytes(fa)
h = s.search(); rows = b"".join(h.rows()); want = open(os.path.join(d, "o.sc"), "rb").read()
ok = rows == want
for q in range(r.nqueries):
    if not np.array_equal(s.query_candidates(q), r.cands(q)): ok = False; print("cands differ", q); break
print("LONG TEST", "OK" if ok else "MISMATCH", len(rows), len(want), "rows", len(r.ints))

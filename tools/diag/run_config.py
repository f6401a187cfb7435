"""Run one BASELINE-style config on the GPU, report time/counters, and check a query sub-range
against the oracle.   python tools/diag/run_config.py N seed [lo hi]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from swiftortho_amd import fsearch, synthprot
from oracle import oracle

N, seed = int(sys.argv[1]), sys.argv[2]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 64)
t = time.time(); fa = synthprot.synthprot(N, 300); print("synth %.1fs %d bytes" % (time.time() - t, len(fa)))
kw = dict(ssd=seed, ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
s = fsearch.Searcher(profile=True, **kw)
t = time.time(); s.load_ref_bytes(fa); s.load_queries_bytes(fa); print("load %.2fs" % (time.time() - t))
REPS = int(os.environ.get("REPS", "2"))
for rep in range(REPS):
    s.reset_counters(); s.drop_index()
    t = time.time(); s.build_index(); ti = time.time() - t
    t = time.time(); h = s.search(); dt = time.time() - t
    c = s.counters()
    print("rep %d index %.3fs search %.3fs rows=%d  %.2f M query-aa/s (index+search)" % (rep, ti, dt, len(h), c["query_aa"] / (dt + ti) / 1e6))
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in c.items()})
    print({k: round(v, 2) for k, v in s.timing().items()})
    if rep + 1 < REPS:
        h.close()
if REPS:
    g = h.array()
    nsub = int(((g["qidx"] >= lo) & (g["qidx"] < hi)).sum())
else:  # REPS=0: only the sampled parity check (index + sub-range search)
    s.build_index()
    nsub = None
d = tempfile.mkdtemp(); p = os.path.join(d, "x.fsa"); open(p, "wb").write(fa)
oracle.build()
t = time.time()
r = oracle.blastp(p, p, os.path.join(d, "o.sc"), ssd=seed, expect=1e-5, v=500, step=1, ht=120000000, chk=50000, st=lo, ed=hi)
print("oracle [%d,%d) %.1fs rows=%d" % (lo, hi, time.time() - t, len(r.ints)))
h2 = s.search(lo, hi); rows = b"".join(h2.rows())
want = open(os.path.join(d, "o.sc"), "rb").read()
print("PARITY", "OK" if rows == want and nsub in (None, len(r.ints)) else "MISMATCH", len(rows), len(want))

"""What the two 30 000-residue singletons of the heterogeneous bench set cost: the same set with and without them, one search step each
(stage clocks on).   python tools/diag/het_giants.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (first: libsohit binds to the same HIP runtime)
from swiftortho_amd import fsearch, synthprot
import bench

fa = synthprot.synthprot(100000, 300, lengths="lognormal")
recs = fa.split(b">")[1:]
small = b"".join(b">" + r for r in recs if len(r) < 20000)
for name, data in (("with giants", fa), ("without", small)):
    s = fsearch.Searcher(device=0, ssd="11111011111", **bench.BASE)
    s.load_ref_bytes(data)
    s.load_queries_bytes(data)
    for k in range(3):
        s.drop_index()
        t = time.perf_counter()
        s.build_index()
        h = s.search()
        n = len(h)
        h.close()
        dt = time.perf_counter() - t
    aa = int(s.query_lengths().sum())
    s.set_profile(True) if hasattr(s, "set_profile") else None
    print("%-12s %d proteins %d aa: %.1f ms per step, %.1f M query-aa/s, %d rows" % (name, len(s.query_lengths()), aa, dt * 1e3, aa / dt / 1e6, n))
    s.close()

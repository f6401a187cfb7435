"""stage laps of the FIRST search of a process against the second (same inputs): where the cold cost of a one-shot find_hit goes.
   python tools/diag/cold_start.py [proteins] [seed pattern]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
from swiftortho_amd import fsearch, synthprot
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ssd = sys.argv[2] if len(sys.argv) > 2 else "11111011111"
fa = synthprot.synthprot(n, 300)
kw = dict(ssd=ssd, nr="AST,CFILMVY,DN,EQ,G,H,KR,P,W", ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
t = time.perf_counter(); s = fsearch.Searcher(**kw); print("create %.3f" % (time.perf_counter() - t))
t = time.perf_counter(); s.load_ref_bytes(fa); s.load_queries_bytes(fa); print("load %.3f" % (time.perf_counter() - t))
for k in range(3):
    s.set_profile(True); s.reset_counters(); s.drop_index()
    t = time.perf_counter(); h = s.search(); dt = time.perf_counter() - t
    tm = s.timing(); c = s.counters()
    print("search %d: %.3f s  index_ms %.1f  stages %s" % (k, dt, c["index_ms"], {a: round(b, 1) for a, b in sorted(tm.items())}))
    t = time.perf_counter(); h.close(); print("  free %.3f" % (time.perf_counter() - t))
t = time.perf_counter(); s.close(); print("close %.3f" % (time.perf_counter() - t))

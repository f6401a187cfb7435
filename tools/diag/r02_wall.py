"""Wall-clock split of a bench step (index build / search_loaded / Python side), profile off.  usage: python tools/diag/r02_wall.py [c3|c2]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from swiftortho_amd import fsearch, synthprot
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
W = bench.WORKLOADS[wl]
print(W)
n_prot, ssd = W[0], W[1]
fa = synthprot.synthprot(n_prot, bench.L_PROT)
s = fsearch.Searcher(device=0, ssd=ssd, **bench.BASE)
s.load_ref_bytes(fa); s.load_queries_bytes(fa)
for it in range(4):
    s.reset_counters()
    t0 = time.perf_counter(); s.drop_index(); s.build_index(); t1 = time.perf_counter()
    hits = s.search(0, n_prot); t2 = time.perf_counter()
    n = len(hits); hits.close(); t3 = time.perf_counter()
    c = s.counters()
    print("iter %d: index %.2f ms  search %.2f ms  close %.2f ms | total_ms %.2f seed %.2f group %.2f phase2 %.2f rows %d" % (
        it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, c["total_ms"], c["seed_ms"], c["group_ms"], c["phase2_ms"], n))

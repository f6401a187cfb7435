"""Does running two half-size searches on two HIP streams (two contexts, two host threads) beat running them back
to back?  Estimates what pipelining the passes of one search over two streams could gain."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from swiftortho_amd import fsearch, synthprot

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
fa = synthprot.synthprot(N, 300)
kw = dict(ssd="111111", ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
S = []
for k in range(P):
    s = fsearch.Searcher(**kw); s.load_ref_bytes(fa); s.load_queries_bytes(fa); s.build_index(); S.append(s)
cuts = [N * k // P for k in range(P + 1)]

def run(s, lo, hi, out, k):
    h = s.search(lo, hi); out[k] = len(h); h.close()

for rep in range(3):
    t = time.time(); h = S[0].search(); n0 = len(h); h.close(); t_full = time.time() - t
    out = [0] * P
    t = time.time()
    for k in range(P): run(S[0], cuts[k], cuts[k + 1], out, k)
    t_seq = time.time() - t
    out2 = [0] * P
    th = [threading.Thread(target=run, args=(S[k], cuts[k], cuts[k + 1], out2, k)) for k in range(P)]
    t = time.time()
    for x in th: x.start()
    for x in th: x.join()
    t_par = time.time() - t
    print("rep %d  full %.1f ms (%d rows)   %d parts back to back %.1f ms   %d parts concurrently %.1f ms (%d rows)"
          % (rep, t_full * 1e3, n0, P, t_seq * 1e3, P, t_par * 1e3, sum(out2)))

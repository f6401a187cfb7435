"""Randomised GPU-vs-oracle differential on the GPU box:  python tools/diag/fuzz_parity.py [cases] [seed]
Draws workload shapes and flag combinations, runs libsohit (HIP) and the oracle on the same input and compares the
formatted rows byte for byte plus the per-query candidate lists.  Exit code 1 on the first mismatch."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SOHIT_KEEP_CANDS"] = "1"
from swiftortho_amd import fsearch, synthprot
from oracle import oracle

oracle.build()
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
AA9 = oracle.AA9
ALPHAS = [AA9, AA9 + "/A,KR,EDNQ,C,G,H,ILVM,FYW,P,ST", "A,C,D,E,F,G,H,I,K,L,M,N,P,Q,R,S,T,V,W,Y"]
SEEDS = ["111111", "1101011", "111111,1101011", "11111011111", "1110111", "11011"]
d = tempfile.mkdtemp()
bad = 0
for case in range(ncase):
    N = int(rng.integers(150, 1600)); L = int(rng.integers(40, 420))
    if os.environ.get("FUZZ_LONG"):  # few, long sequences: tiled alignments, wide position fields
        N = int(rng.integers(20, 120)); L = int(rng.integers(500, 6000))
    uniform = rng.random() < 0.25
    if os.environ.get("FUZZ_HET") and rng.random() < 0.7:  # log-normal lengths with a tail and one 30 000-residue protein: every length class in one batch
        N = int(rng.integers(120, 900))
        fa = synthprot.synthprot(N, seed=int(rng.integers(1, 1 << 30)), lengths="lognormal")
    else:
        fa = (synthprot.uniform_proteins if uniform else synthprot.synthprot)(N, L, int(rng.integers(1, 1 << 30)))
    if rng.random() < 0.35:  # odd residues: gap / stop characters, masked and ambiguous letters, lower case, a '\r'
        lines = fa.split(b"\n")
        odd = b"-*xXUuBZJO.a" + b"lkde\r"
        for li, ln in enumerate(lines):
            if ln and not ln.startswith(b">") and rng.random() < 0.2:
                bb = bytearray(ln)
                for pos in rng.integers(0, len(bb), size=int(rng.integers(1, 6))):
                    bb[int(pos)] = odd[int(rng.integers(0, len(odd)))]
                lines[li] = bytes(bb)
        fa = b"\n".join(lines)
    kw = dict(ssd=str(rng.choice(SEEDS)), nr=str(rng.choice(ALPHAS)), ht=int(rng.choice([50021, 1000003, 15000017, 120000000])),
              chk=int(rng.choice([50000, N // 3 + 1, 97])), step=int(rng.choice([1, 1, 2, 4])), v=int(rng.choice([500, 50, 5, 1200])),
              expect=float(rng.choice([1e-5, 1e-3, 10.0])), flt=str(rng.choice(["T", "T", "F"])),
              thr=int(rng.choice([-1, -1, 3, 40])), max_miss=float(rng.choice([1e-3, 0.5])))
    if kw["nr"].count(",") > 15 and kw["ssd"] in ("11011",):
        kw["ssd"] = "1111111"   # weight-4 seeds on a 20-letter alphabet explode the hit lists of the CPU oracle
    if rng.random() < 0.2:  # corner values
        kw["chk"] = int(rng.choice([1, 2, 5])); kw["step"] = int(rng.choice([7, 50]))
    rst, red = (-1, -1) if rng.random() < 0.7 else (int(rng.integers(0, N // 2)), int(rng.integers(N // 2, N + 50)))
    lo = int(rng.integers(0, N // 2)); hi = int(min(N, lo + rng.integers(20, 120)))
    os.environ["SOHIT_BATCH"] = str(int(rng.choice([16384, 37])))
    os.environ["SOHIT_MAX_HITS"] = str(int(rng.choice([1 << 30, 50000])))
    if os.environ.get("FUZZ_ONLY") and case != int(os.environ["FUZZ_ONLY"]):
        continue  # (the draws above keep the random stream aligned)
    p = os.path.join(d, "x.fsa"); open(p, "wb").write(fa)
    out = os.path.join(d, "o.sc")
    r = oracle.blastp(p, p, out, ssd=kw["ssd"], nr=kw["nr"], expect=kw["expect"], v=kw["v"], step=kw["step"], flt=kw["flt"], ht=kw["ht"],
                      chk=kw["chk"], st=lo, ed=hi, thr=kw["thr"], max_miss=kw["max_miss"], rst=rst, red=red)
    s = fsearch.Searcher(**kw); s.load_ref_bytes(fa, rst, red); s.load_queries_bytes(fa)
    h = s.search(lo, hi); rows = b"".join(h.rows()); want = open(out, "rb").read()
    ok = rows == want
    for qrel in range(r.nqueries):
        if not np.array_equal(s.query_candidates(lo + qrel), r.cands(qrel)):
            ok = False
            print("  candidates of query", lo + qrel, "differ")
            a, b = s.query_candidates(lo + qrel), r.cands(qrel)
            print("   gpu %d oracle %d candidates" % (len(a), len(b)))
            sa, sb = set(map(tuple, a.tolist())), set(map(tuple, b.tolist()))
            print("   only gpu:", sorted(sa - sb)[:6], " only oracle:", sorted(sb - sa)[:6])
            if sa == sb:
                k = next(i for i in range(len(a)) if tuple(a[i]) != tuple(b[i]))
                print("   same set, order differs from position", k, a[k:k + 3].tolist(), b[k:k + 3].tolist())
            break
    print("case %2d %s N=%d L=%d %s rows=%d  %s" % (case, "ok  " if ok else "FAIL", N, L, "uniform" if uniform else "families", len(r.ints),
                                                    {k: kw[k] for k in ("ssd", "ht", "chk", "step", "v", "expect", "flt", "thr", "max_miss")}),
          "alpha=%d" % ALPHAS.index(kw["nr"]), "ref", (rst, red), "batch", os.environ["SOHIT_BATCH"], "maxhits", os.environ["SOHIT_MAX_HITS"], flush=True)
    h.close(); s.close()
    if not ok:
        bad += 1
        open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "fuzz_fail_%d.fsa" % case), "wb").write(fa)
        break
sys.exit(1 if bad else 0)

"""where find_cluster's time goes on config 5: python tools/diag/mclprof.py prepare <dir>  /  ... run <dir>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mode, d = sys.argv[1], sys.argv[2]
if mode == "prepare":
    from swiftortho_amd import pipeline, synthprot
    os.makedirs(d, exist_ok=True)
    p = os.path.join(d, "x.fsa")
    open(p, "wb").write(synthprot.synthprot(100000, 300))
    lines, tm = pipeline.orthology_from_search(p, ssd="11111011111", nr="aa9", ht=120000000, chk=50000, step=1, v=500, expect=1e-5, flt="T")
    open(os.path.join(d, "x.opc"), "wb").write(b"".join(l + b"\n" for l in lines))
    print("relations", len(lines))
else:
    from swiftortho_amd import find_cluster as fc
    data = open(os.path.join(d, "x.opc"), "rb").read()
    calls = []
    def timed(ip, ix, dv, infl, **k):
        t = time.time(); r = fc.device_mcl(ip, ix, dv, infl, **k); calls.append((len(ip) - 1, len(ix), time.time() - t)); return r
    t = time.time(); g = fc.cnc(data, 1.5, mcl=timed); tt = time.time() - t
    print("cnc %.2f s, groups %d, mcl calls %s" % (tt, len(g), [(n, nnz, round(s, 3)) for n, nnz, s in calls]))
    t = time.time(); g = fc.cnc(data, 1.5, mcl=timed); print("second run %.2f s" % (time.time() - t), [(n, nnz, round(s, 3)) for n, nnz, s in calls[len(calls)//2:]])

"""Container-only randomised differential: the REAL reference (CPython-hosted, refload.py) against oracle/sohit_cpu on
small random inputs and flag combinations -- odd residues, -L/-U and -l/-u ranges, tiny chunks, several seeds and
alphabets, steps, -v / -e / -t / -m / -F corners.  Pins the oracle beyond the committed goldens; CPU only.

    python tools/refharness/fuzz_oracle.py [cases] [rng seed]

Exit code 0 == every case byte-identical (all 16 columns)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refload  # noqa: E402
from oracle import oracle  # noqa: E402
from swiftortho_amd import synthprot  # noqa: E402

AA9 = oracle.AA9
ALPHAS = [AA9, AA9 + "/A,KR,EDNQ,C,G,H,ILVM,FYW,P,ST", "A,C,D,E,F,G,H,I,K,L,M,N,P,Q,R,S,T,V,W,Y"]
SEEDS = ["111111", "1101011", "111111,1101011", "11111011111", "1110111", "11111"]


def main():
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    oracle.build()
    m = refload.load()
    tmp = tempfile.mkdtemp(prefix="fuzzoracle_")
    bad = 0
    for case in range(ncase):
        N = int(rng.integers(12, 70))
        L = int(rng.integers(40, 220))
        uniform = rng.random() < 0.2
        fa = (synthprot.uniform_proteins if uniform else synthprot.synthprot)(N, L, int(rng.integers(1, 1 << 30)))
        if rng.random() < 0.5:
            lines = fa.split(b"\n")
            odd = b"-*xXUuBZJO.a" + b"lkde"
            for li, ln in enumerate(lines):
                if ln and not ln.startswith(b">") and rng.random() < 0.3:
                    bb = bytearray(ln)
                    for pos in rng.integers(0, len(bb), size=int(rng.integers(1, 6))):
                        bb[int(pos)] = odd[int(rng.integers(0, len(odd)))]
                    lines[li] = bytes(bb)
            fa = b"\n".join(lines)
        nseq = fa.count(b">")
        flags = ["-s", str(rng.choice(SEEDS)), "-r", str(rng.choice(ALPHAS)), "-M", str(int(rng.choice([13, 5003, 1000003]))),
                 "-c", str(int(rng.choice([50000, 1, 3, nseq // 3 + 1]))), "-j", str(int(rng.choice([1, 1, 2, 5, 40]))),
                 "-v", str(int(rng.choice([500, 1, 3, 50]))), "-e", str(rng.choice(["1e-5", "1e-3", "10", "1e300", "0"])),
                 "-F", str(rng.choice(["T", "T", "F"])), "-t", str(int(rng.choice([-1, -1, 0, 1, 30]))),
                 "-m", str(rng.choice(["1e-3", "0.5", "0", "3"]))]
        if rng.random() < 0.3:
            a, b = int(rng.integers(0, nseq)), int(rng.integers(0, nseq + 5))
            flags += ["-l", str(min(a, b)), "-u", str(max(a, b))]
        if rng.random() < 0.3:
            a, b = int(rng.integers(0, nseq // 2)), int(rng.integers(nseq // 2, nseq + 5))
            flags += ["-L", str(a), "-U", str(b)]
        p = os.path.join(tmp, "x.fsa")
        open(p, "wb").write(fa)
        ro, oo = os.path.join(tmp, "ref.sc"), os.path.join(tmp, "ora.sc")
        for f in (ro, oo):
            if os.path.exists(f):
                os.remove(f)
        try:
            m.entry_point(["fsearch", "-p", "blastp", "-i", p, "-d", p, "-o", ro, "-T", tmp] + flags)
        except Exception as e:  # inputs the reference itself cannot process are not parity cases
            print("case %3d skipped (reference raised %s: %s)  %s" % (case, type(e).__name__, e, " ".join(flags)), flush=True)
            continue
        subprocess.run([oracle.EXE, "-p", "blastp", "-i", p, "-d", p, "-o", oo, "-T", tmp] + flags, check=True)
        r = open(ro, "rb").read() if os.path.exists(ro) else b""
        o = open(oo, "rb").read() if os.path.exists(oo) else b""
        ok = r == o
        print("case %3d %s N=%d L=%d rows=%d  %s" % (case, "ok  " if ok else "FAIL", nseq, L, r.count(b"\n"), " ".join(flags)), flush=True)
        if not ok:
            bad += 1
            keep = os.path.join(ROOT, "gpurun_out", "fuzz_oracle_fail_%d.fsa" % case)
            os.makedirs(os.path.dirname(keep), exist_ok=True)
            open(keep, "wb").write(fa)
            ra, oa = r.split(b"\n"), o.split(b"\n")
            for i in range(max(len(ra), len(oa))):
                x = ra[i] if i < len(ra) else b"<none>"
                y = oa[i] if i < len(oa) else b"<none>"
                if x != y:
                    print("   row %d\n    ref: %r\n    ora: %r" % (i, x, y))
                    break
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

"""Container-only differential check: REAL reference (CPython-hosted, refload.py)
versus oracle/sohit_cpu, end to end and stage by stage, on seeded random inputs.

    python tools/refharness/diff_oracle.py [--n 60] [--seeds 111111] [--M 1000003] [--c 50000] [--uniform]

Exit code 0 == byte-identical .sc output (all 16 columns).
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refload  # noqa: E402
from oracle import oracle  # noqa: E402
from swiftortho_amd import synthprot  # noqa: E402


def run_reference(m, qry, ref, out, args, tmpdir):
    argv = ["fsearch", "-p", "blastp", "-i", qry, "-d", ref, "-o", out, "-T", tmpdir] + args
    m.entry_point(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=60)
    ap.add_argument("--L", type=int, default=120)
    ap.add_argument("--seeds", default="111111")
    ap.add_argument("--r", default=oracle.AA9)
    ap.add_argument("--M", type=int, default=1000003)
    ap.add_argument("--c", type=int, default=50000)
    ap.add_argument("--e", default="1e-5")
    ap.add_argument("--v", default="500")
    ap.add_argument("--j", default="1")
    ap.add_argument("--F", default="T")
    ap.add_argument("--rng", type=int, default=7)
    ap.add_argument("--uniform", action="store_true")
    ap.add_argument("--keep", default="")
    a = ap.parse_args()

    m = refload.load()
    tmp = tempfile.mkdtemp(prefix="difforacle_")
    fa = os.path.join(tmp, "x.fsa")
    data = synthprot.uniform_proteins(a.n, a.L, a.rng) if a.uniform else synthprot.synthprot(a.n, a.L, a.rng)
    open(fa, "wb").write(data)
    flags = ["-e", a.e, "-v", a.v, "-s", a.seeds, "-r", a.r, "-M", str(a.M), "-c", str(a.c), "-j", a.j, "-F", a.F]
    t0 = time.time()
    run_reference(m, fa, fa, os.path.join(tmp, "ref.sc"), flags, tmp)
    t1 = time.time()
    subprocess.run([oracle.EXE, "-p", "blastp", "-i", fa, "-d", fa, "-o", os.path.join(tmp, "ora.sc"), "-T", tmp] + flags,
                   check=True)
    t2 = time.time()
    r = open(os.path.join(tmp, "ref.sc"), "rb").read()
    o = open(os.path.join(tmp, "ora.sc"), "rb").read()
    print("reference %.1fs (%d rows)  oracle %.2fs (%d rows)" % (t1 - t0, r.count(b"\n"), t2 - t1, o.count(b"\n")))
    if a.keep:
        import shutil
        shutil.copytree(tmp, a.keep, dirs_exist_ok=True)
    if r != o:
        rl, ol = r.split(b"\n"), o.split(b"\n")
        for i in range(max(len(rl), len(ol))):
            x = rl[i] if i < len(rl) else b"<none>"
            y = ol[i] if i < len(ol) else b"<none>"
            if x != y:
                print("first diff at row", i)
                print(" ref:", x.decode("latin-1"))
                print(" ora:", y.decode("latin-1"))
                break
        sys.exit(1)
    print("IDENTICAL")


if __name__ == "__main__":
    main()

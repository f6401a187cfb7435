"""Container-only: goldens for the MCL clustering stage.

    python tools/refharness/make_cluster_goldens.py [--force]

Runs the REAL reference script /root/reference/bin/find_cluster.py -a mcl (numpy + scipy + networkx are in the
image; numba and cffi are not, so tools/refharness/fcshim/ supplies no-op stand-ins for those two imports) on
.orth files and stores its stdout.  Fixtures: tests/golden/clu_<name>.orth (input; for the orth_* goldens the
existing expected-output file is the input), clu_<name>.I<inflation>.mcl (expected stdout), clu_<name>.json.
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REFERENCE = os.environ.get("SWIFTORTHO_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
FORCE = "--force" in sys.argv


def run_ref_find_cluster(orth_path, flags):
    with tempfile.TemporaryDirectory(prefix="clu_") as d:
        local = os.path.join(d, "in.orth")
        open(local, "wb").write(open(orth_path, "rb").read())
        env = dict(os.environ, LC_ALL="C", PYTHONPATH=os.path.join(HERE, "fcshim"))
        r = subprocess.run([sys.executable, os.path.join(REFERENCE, "bin", "find_cluster.py"), "-i", local] + flags, cwd=d,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        if r.returncode != 0:
            raise RuntimeError(r.stderr.decode()[-3000:])
        return r.stdout


def make(name, orth_file, inflations=("1.5", "2.0")):
    meta_path = os.path.join(GOLD, "clu_%s.json" % name)
    if os.path.isfile(meta_path) and not FORCE:
        print(name, "exists, skipped")
        return
    meta = {"input": os.path.basename(orth_file), "variants": {}}
    for I in inflations:
        out = run_ref_find_cluster(orth_file, ["-a", "mcl", "-I", I])
        open(os.path.join(GOLD, "clu_%s.I%s.mcl" % (name, I)), "wb").write(out)
        meta["variants"]["I" + I] = ["-a", "mcl", "-I", I]
        ids = set(open(orth_file).read().replace("\n", "\t").split("\t")[1::4]) | set(open(orth_file).read().replace("\n", "\t").split("\t")[2::4])
        print(name, "I =", I, "clusters", out.count(b"\n"), "genes clustered", len(out.split()), "of", len(ids))
    json.dump(meta, open(meta_path, "w"), indent=1)


def main():
    for n in ("taxa5", "taxa3_dense", "taxa4_colon", "toy_default"):
        make(n, os.path.join(GOLD, "orth_%s.default.orth" % n))
    make("taxa5_bsr", os.path.join(GOLD, "orth_taxa5.bsr.orth"), inflations=("1.5",))
    # a larger graph: 8 taxa, 260 families with many in-paralogs (fused and split clusters)
    big = os.path.join(GOLD, "clu_taxa8_big.orth")
    if FORCE or not os.path.isfile(big):
        import make_orth_goldens as mo
        from oracle import oracle
        oracle.build()
        with tempfile.TemporaryDirectory() as d:
            fa, sc = os.path.join(d, "x.fsa"), os.path.join(d, "x.sc")
            open(fa, "wb").write(mo.taxa_fasta(260, 8, 100, 21, p_copy=0.85, p_dup=0.45))
            subprocess.run([oracle.EXE, "-p", "blastp", "-i", fa, "-d", fa, "-o", sc, "-e", "1e-5", "-v", "500", "-j", "1", "-F", "T", "-s", "111111",
                            "-M", "1000003", "-c", "50000"], check=True, stderr=subprocess.DEVNULL)
            open(big, "wb").write(mo.run_ref_find_orth(sc, []))
    make("taxa8_big", big)


if __name__ == "__main__":
    main()

"""BASELINE config 5 (100k-protein find_hit -> find_orth -> find_cluster -a mcl -I 1.5): stage wall times on the GPU box, the three
drop-in CLIs and the in-process flow that hands hit RECORDS to find_orth.   python tools/diag/c5_stages.py [proteins]"""
import hashlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from swiftortho_amd import find_cluster, pipeline, synthprot

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
meta = json.load(open(os.path.join(ROOT, "tests", "golden", "pipe_c3.json" if n == 100000 else "pipe_c2.json")))
d = dict(zip(meta["find_hit_flags"][0::2], meta["find_hit_flags"][1::2]))
t = time.time(); fa = synthprot.synthprot(n, 300); print("synthprot %.1f s" % (time.time() - t))
tmp = tempfile.mkdtemp()
p, sc, op = os.path.join(tmp, "x.fsa"), os.path.join(tmp, "x.sc"), os.path.join(tmp, "x.opc")
open(p, "wb").write(fa)
py = sys.executable
print("== command-line chain (text files between the stages) ==")
t = time.time()
subprocess.run([py, os.path.join(ROOT, "bin", "find_hit.py"), "-p", "blastp", "-i", p, "-d", p, "-o", sc, "-a", "1", "-e", d["-e"], "-v", d["-v"], "-j", d["-j"], "-F", d["-F"],
                "-s", d["-s"], "-r", "aa9", "-M", d["-M"], "-c", d["-c"]], check=True)
t1 = time.time() - t; print("find_hit.py (process start, FASTA parse, GPU search, %d rows of text written): %.2f s" % (open(sc).read().count("\n"), t1))
t = time.time()
orth = subprocess.run([py, os.path.join(ROOT, "bin", "find_orth.py"), "-i", sc], capture_output=True, check=True).stdout
open(op, "wb").write(orth)
t2 = time.time() - t; print("find_orth.py (library tokeniser + columnar stage, %d relations): %.2f s" % (orth.count(b"\n"), t2))
t = time.time()
grp = subprocess.run([py, os.path.join(ROOT, "bin", "find_cluster.py"), "-i", op, "-a", "mcl", "-I", "1.5"], capture_output=True, check=True).stdout
t3 = time.time() - t; print("find_cluster.py -a mcl (columnar graph bookkeeping + Markov loop on the GPU, %d groups): %.2f s" % (grp.count(b"\n"), t3))
print("orth md5 ok:", hashlib.md5(orth).hexdigest() == meta["orth_md5"], " groups sha256 ok:", hashlib.sha256(grp).hexdigest() == meta["groups_text_sha256"])
print("== in-process: hit records handed to find_orth, no .sc parsed ==")
lines, tm = pipeline.orthology_from_search(p, ssd=d["-s"], nr=d["-r"], ht=int(d["-M"]), chk=int(d["-c"]), step=int(d["-j"]), v=int(d["-v"]), expect=float(d["-e"]), flt=d["-F"])
print({k: round(v, 3) if isinstance(v, float) else v for k, v in tm.items()})
print("relations identical:", b"".join(l + b"\n" for l in lines) == orth)
t = time.time()
groups = find_cluster.cnc([l.decode() + "\n" for l in lines], 1.5)
print("find_cluster.cnc in-process: %.2f s, %d groups" % (time.time() - t, len(groups)))

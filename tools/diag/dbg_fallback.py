import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["SOHIT_BUCKET_MIN"]="0"; os.environ["SOHIT_BUCKET_AVG"]="64"; os.environ["SOHIT_DEBUG"]="1"
from swiftortho_amd import fsearch, synthprot
fa = synthprot.synthprot(24, 5200, 1258)
s = fsearch.Searcher(ssd="111111", ht=1000003, chk=50000, step=1, v=500, expect=1e-5, flt="F")
s.load_ref_bytes(fa); s.load_queries_bytes(fa)
h = s.search(); print("rows", len(h))

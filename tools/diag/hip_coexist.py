"""Diagnostic: can torch (bundled ROCm runtime) and libsohit.so (system ROCm) share a process?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
order = sys.argv[1] if len(sys.argv) > 1 else "torch_first"

def maps():
    seen = set()
    for line in open("/proc/self/maps"):
        p = line.split()[-1]
        if ("amdhip64" in p or "hsa-runtime" in p) and p not in seen:
            seen.add(p); print("  mapped:", p)

def mk():
    from swiftortho_amd import fsearch
    try:
        s = fsearch.Searcher(ht=1000003)
        print("  so_create OK"); s.close()
    except Exception as e:
        print("  so_create FAILED:", e)

if order == "torch_first":
    import torch
    print("torch", torch.__version__, "cuda avail", torch.cuda.is_available(), "count", torch.cuda.device_count())
    x = torch.ones(4, device="cuda"); print("  torch tensor ok", float(x.sum()))
    mk(); maps()
else:
    mk()
    import torch
    print("torch cuda avail", torch.cuda.is_available())
    try:
        x = torch.ones(4, device="cuda"); print("  torch tensor ok", float(x.sum()))
    except Exception as e:
        print("  torch FAILED", e)
    mk(); maps()

"""where the 0.25 s of so_create go: library load, HIP runtime start (first API call), the context itself.   python tools/diag/create_cost.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("SOHIT_TORCH_PRELOAD", sys.argv[1] if len(sys.argv) > 1 else "0")
t0 = time.perf_counter()
from swiftortho_amd import _lib
L = _lib.load()
t1 = time.perf_counter()
hip = C.CDLL("libamdhip64.so")
n = C.c_int(0)
hip.hipGetDeviceCount(C.byref(n))
t2 = time.perf_counter()
hip.hipSetDevice(0)
hip.hipFree(None)
t3 = time.perf_counter()
from swiftortho_amd import fsearch
s = fsearch.Searcher(ssd="11111011111", ht=120000000)
t4 = time.perf_counter()
print("preload=%s  load libsohit %.3f  hipGetDeviceCount (runtime start) %.3f  hipSetDevice+hipFree(0) (context) %.3f  so_create after that %.3f" % (
    os.environ["SOHIT_TORCH_PRELOAD"], t1 - t0, t2 - t1, t3 - t2, t4 - t3))

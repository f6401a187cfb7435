"""Build libsohit.so (HIP kernels + C++ host, gfx950 only) in-tree with hipcc.

    python -m swiftortho_amd.build [--force]

The shared library lands next to this file (swiftortho_amd/libsohit.so) so that it
travels with the repository snapshot to the GPU box; it is git-ignored.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libsohit.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

SOURCES = ["k_util.hip", "k_sort.hip", "k_sortseg.hip", "k_prep.hip", "k_index.hip", "k_ixsort.hip", "k_seed.hip", "k_group.hip", "k_ungap1.hip", "k_ungapq.hip", "k_bucket.hip", "k_align.hip", "k_align16.hip", "k_alignl.hip", "k_phase2.hip", "mcl.hip",
           "tsv.hip", "host_load.hip", "host_index.hip", "host_seed.hip", "host_phase2.hip", "host_search.hip", "host_abi.hip"]
# -ffp-contract=off: host-side SEG/threshold arithmetic must round exactly like the reference's
# (no fused multiply-add), and device fp64 compares stay IEEE.
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-I", CSRC, "-I", INCLUDE] + os.environ.get("SOHIT_CXXFLAGS", "").split()   # (diagnostic builds: -DUG_STATS)


def _deps_mtime():
    m = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith((".h", ".hpp")):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if (not force and os.path.isfile(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), _deps_mtime())):
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stdout[-6000:]))
    if r.stdout.strip():
        sys.stderr.write(r.stdout)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.isfile(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout[-4000:])
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

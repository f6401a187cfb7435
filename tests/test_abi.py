"""CPU-only: the C-ABI library loads and exports every symbol include/sohit.h declares; the
product refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from swiftortho_amd import build
    build.build(verbose=False)
    from swiftortho_amd import _lib
    return _lib


def header_symbols():
    h = open(os.path.join(ROOT, "include", "sohit.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(so_[a-z0-9_]+)\s*\(", h)))


def test_every_declared_symbol_is_exported(lib):
    L = C.CDLL(lib.LIBPATH)
    declared = header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), "libsohit.so does not export %s" % name
    assert sorted(lib.EXPORTS) == declared, "swiftortho_amd/_lib.py EXPORTS out of sync with include/sohit.h"


def test_struct_layouts_match_header(lib):
    # so_hit: 2 x i64, 2 x f64, 12 x i32 ; so_params: 2 ptr, 5 x i64, 2 x f64, 2 x i32
    assert C.sizeof(lib.SoHit) == 80
    assert C.sizeof(lib.SoParams) == 2 * C.sizeof(C.c_void_p) + 5 * 8 + 2 * 8 + 2 * 4
    assert lib.load().so_abi_version() == 3


def test_no_cpu_fallback(lib):
    """Without a usable HIP device so_create must fail with a message -- never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from swiftortho_amd import fsearch
    with pytest.raises(fsearch.SohitError) as e:
        fsearch.Searcher(ht=1000003)
    assert "no HIP device" in str(e.value) or "HIP" in str(e.value)


def test_bad_parameters_are_reported_not_aborted(lib):
    L = lib.load()
    p = lib.SoParams(b"1111111", b"AST,CFILMVY,DN,EQ,G,H,KR,P,W", -1, 50000, 1, 500, -1, 1e-5, 1e-3, 1, 0)  # -M < 1, odd weight: refused
    assert not L.so_create(0, C.byref(p))
    assert L.so_last_error(None)
    assert L.so_create(0, None) is None or not L.so_create(0, None)
    assert L.so_num_queries(None) == -1 and L.so_get_counters(None, None) != 0


def test_product_never_imports_the_oracle():
    """oracle/ and tests/mcl_scipy_oracle.py are test infrastructure: nothing under swiftortho_amd/ or bin/ may reference them (nor scipy,
    which only the MCL oracle uses)."""
    bad = []
    for base in ("swiftortho_amd", "bin"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            if "_build" in d or "__pycache__" in d:
                continue
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    txt = open(os.path.join(d, f), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b|liboracle|sohit_cpu|mcl_scipy_oracle|^\s*(from|import)\s+scipy", txt, flags=re.M):
                        if f == "host.h" and "oracle/" in txt:
                            # the header comment forbids it; make sure there is no include/link
                            if not re.search(r"#include\s+\"[^\"]*oracle", txt):
                                continue
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def test_mcl_fails_loudly_without_a_gpu(lib):
    """the Markov loop has no CPU path either: so_mcl reports the missing device"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    from swiftortho_amd import find_cluster as fc
    with pytest.raises(RuntimeError) as e:
        fc.device_mcl(np.array([0, 1, 2, 2]), np.array([0, 1], dtype=np.int32), np.array([1, 1], dtype=np.float32), 1.5)
    assert "HIP" in str(e.value)

"""ctypes binding of libsohit.so (include/sohit.h).  No fallback: if the HIP library is
missing or no GPU is usable, importing/creating a context raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(HERE, "libsohit.so")


class SoParams(C.Structure):
    _fields_ = [("seeds", C.c_char_p), ("alphabet", C.c_char_p), ("nc", C.c_int64), ("chunk", C.c_int64), ("step", C.c_int64),
                ("max_hits", C.c_int64), ("thr", C.c_int64), ("expect", C.c_double), ("max_miss", C.c_double),
                ("filter", C.c_int32), ("profile", C.c_int32)]


class SoHit(C.Structure):
    _fields_ = [("qidx", C.c_int64), ("sidx", C.c_int64), ("identity", C.c_double), ("evalue", C.c_double), ("aln", C.c_int32),
                ("mis", C.c_int32), ("gap", C.c_int32), ("qst", C.c_int32), ("qed", C.c_int32), ("sst", C.c_int32),
                ("sed", C.c_int32), ("bit", C.c_int32), ("qlen", C.c_int32), ("slen", C.c_int32), ("matches", C.c_int32),
                ("ungapped", C.c_int32)]


class SoCounters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_queries", "query_aa", "ref_seqs", "ref_aa", "n_chunks", "seed_windows", "seed_hits",
                                         "groups", "candidates", "alignments", "cells", "rows", "index_entries")] + \
               [("lookup_launches", C.c_int64), ("lookup_ms", C.c_double), ("lookup_bytes", C.c_int64),
                ("bounds_launches", C.c_int64), ("bounds_ms", C.c_double), ("bounds_bytes", C.c_int64),
                ("align_launches", C.c_int64), ("align_ms", C.c_double),
                ("index_ms", C.c_double), ("seed_ms", C.c_double), ("group_ms", C.c_double), ("phase2_ms", C.c_double),
                ("total_ms", C.c_double), ("count_launches", C.c_int64), ("count_ms", C.c_double),
                ("hits_bucketed", C.c_int64), ("align_wide", C.c_int64), ("cells_wide", C.c_int64), ("seed_passes", C.c_int64),
                ("ungap_steps", C.c_int64), ("groups_single", C.c_int64), ("groups_chain", C.c_int64),
                ("bgroup_launches", C.c_int64), ("bgroup_ms", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/sohit.h declares
EXPORTS = ["so_abi_version", "so_set_option", "so_create", "so_destroy", "so_last_error", "so_load_ref", "so_load_ref_mem", "so_build_index", "so_drop_index", "so_load_index",
           "so_load_queries", "so_load_queries_mem", "so_num_queries", "so_num_refs", "so_query_len", "so_search_loaded",
           "so_search", "so_free_hits", "so_write_sc", "so_format_hit", "so_get_counters", "so_reset_counters", "so_timing_report",
           "so_chunk_threshold", "so_chunk_entries", "so_chunk_download", "so_masked_query", "so_query_candidates", "so_set_profile",
           "so_bucket_count", "so_ref_len", "so_search_device", "so_device_hits_copy", "so_query_work", "so_mcl", "so_mcl_free",
           "so_mcl_last_error", "so_tsv_lines", "so_tsv_scan", "so_tsv_codes", "so_format_pairs", "so_py_repr", "so_fmt_rows"]


class SoMclResult(C.Structure):
    _fields_ = [("n", C.c_int64), ("nnz", C.c_int64), ("rounds", C.c_int32), ("converged", C.c_int32), ("indptr", C.POINTER(C.c_int64)),
                ("indices", C.POINTER(C.c_int32)), ("data", C.POINTER(C.c_float))]

_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP
    runtimes in one process cannot both open the GPU, so whichever of torch / libsohit loads
    second would fail.  Loading torch's copy first (by path, without importing torch) makes the
    dynamic linker bind libsohit's DT_NEEDED libamdhip64.so.7 to it, and a later `import torch`
    reuses the same mapping.  Without torch installed this is a no-op (system ROCm is used)."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("SOHIT_TORCH_PRELOAD") == "0":   # ("0": a process that will never import torch -- the one-GPU CLI)
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.isfile(p):
        try:
            C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libsohit.so (never builds, never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    _preload_torch_hip_runtime()
    if not os.path.isfile(LIBPATH):
        raise ImportError("libsohit.so is missing (%s): run `python -m swiftortho_amd.build` -- there is no CPU fallback" % LIBPATH)
    L = C.CDLL(LIBPATH)
    vp, i64, cp = C.c_void_p, C.c_int64, C.c_char_p
    L.so_abi_version.restype = C.c_int
    L.so_create.restype = vp
    L.so_create.argtypes = [C.c_int, C.POINTER(SoParams)]
    L.so_destroy.argtypes = [vp]
    L.so_last_error.restype = cp
    L.so_last_error.argtypes = [vp]
    L.so_load_ref.argtypes = [vp, cp, i64, i64]
    L.so_load_ref_mem.argtypes = [vp, cp, i64, i64, i64]
    L.so_build_index.argtypes = [vp]
    L.so_drop_index.argtypes = [vp]
    L.so_load_index.argtypes = [vp, cp]
    L.so_load_queries.argtypes = [vp, cp]
    L.so_load_queries_mem.argtypes = [vp, cp, i64]
    for f in ("so_num_queries", "so_num_refs"):
        getattr(L, f).restype = i64
        getattr(L, f).argtypes = [vp]
    L.so_query_len.restype = i64
    L.so_query_len.argtypes = [vp, i64]
    L.so_ref_len.restype = i64
    L.so_ref_len.argtypes = [vp, i64]
    L.so_search_loaded.argtypes = [vp, i64, i64, C.POINTER(C.POINTER(SoHit)), C.POINTER(i64)]
    L.so_search.argtypes = [vp, cp, i64, i64, C.POINTER(C.POINTER(SoHit)), C.POINTER(i64)]
    L.so_free_hits.argtypes = [C.POINTER(SoHit)]
    L.so_search_device.argtypes = [vp, i64, i64, C.POINTER(vp), C.POINTER(i64)]
    L.so_device_hits_copy.argtypes = [vp, vp, i64]
    L.so_query_work.argtypes = [vp, i64, i64, vp]
    L.so_write_sc.argtypes = [vp, C.POINTER(SoHit), i64, cp, cp]
    L.so_format_hit.restype = i64
    L.so_format_hit.argtypes = [vp, C.POINTER(SoHit), cp, i64]
    L.so_get_counters.argtypes = [vp, C.POINTER(SoCounters)]
    L.so_set_profile.argtypes = [vp, C.c_int]
    L.so_bucket_count.restype = i64
    L.so_bucket_count.argtypes = [vp]
    L.so_reset_counters.argtypes = [vp]
    L.so_timing_report.restype = i64
    L.so_timing_report.argtypes = [vp, cp, i64]
    L.so_tsv_lines.restype = i64
    L.so_tsv_lines.argtypes = [vp, i64, vp, i64]
    L.so_tsv_scan.argtypes = [vp, i64, vp, i64, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    L.so_tsv_codes.restype = i64
    L.so_tsv_codes.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64]
    L.so_format_pairs.restype = i64
    L.so_format_pairs.argtypes = [cp, C.c_int32, vp, vp, vp, vp, vp, i64, vp, i64]
    L.so_py_repr.restype = i64
    L.so_py_repr.argtypes = [vp, i64, vp, i64]
    L.so_fmt_rows.restype = i64
    L.so_fmt_rows.argtypes = [vp, i64, vp, i64]
    for f in ("so_chunk_threshold", "so_chunk_entries"):
        getattr(L, f).restype = i64
        getattr(L, f).argtypes = [vp, i64]
    L.so_chunk_download.argtypes = [vp, i64, vp, vp]
    L.so_masked_query.restype = i64
    L.so_masked_query.argtypes = [vp, i64, cp, i64]
    L.so_query_candidates.restype = i64
    L.so_query_candidates.argtypes = [vp, i64, vp, i64]
    L.so_mcl.argtypes = [C.c_int, i64, vp, vp, vp, C.c_double, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.POINTER(SoMclResult)]
    L.so_mcl_free.argtypes = [C.POINTER(SoMclResult)]
    L.so_mcl_last_error.restype = cp
    L.so_set_option.argtypes = [vp, cp, cp]
    if L.so_abi_version() != 3:
        raise ImportError("libsohit.so ABI version mismatch")
    _lib = L
    return L

"""ctypes binding of oracle/liboracle.so (CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
EXE = os.path.join(HERE, "sohit_cpu")
AA9 = "AST,CFILMVY,DN,EQ,G,H,KR,P,W"
AA20 = "A,S,T,C,F,I,L,M,V,Y,D,N,E,Q,G,H,K,R,P,W"


def build(force=False):
    src = os.path.join(HERE, "sohit_cpu.cpp")
    stale = (not os.path.isfile(LIB) or not os.path.isfile(EXE)
             or os.path.getmtime(LIB) < os.path.getmtime(src))
    if force or stale:
        subprocess.run(["make", "-C", HERE, "-B", "all"], check=True, stdout=subprocess.DEVNULL)


_lib = None
i64 = C.c_int64
p_i64 = C.POINTER(C.c_int64)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.oc_b62.restype = C.c_int
        L.oc_spseeds.restype = i64
        L.oc_spseeds.argtypes = [C.c_char_p, i64, C.c_int, C.c_char_p, C.c_char_p, i64, C.c_void_p, C.c_void_p]
        L.oc_seg.argtypes = [C.c_char_p, i64, C.c_char_p]
        L.oc_qsort.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oc_ungap.argtypes = [C.c_char_p, i64, C.c_char_p, i64, i64, i64, i64, i64, C.c_void_p]
        L.oc_ungap_chain.argtypes = [C.c_char_p, i64, C.c_char_p, i64, C.c_void_p, C.c_int, C.c_void_p]
        L.oc_kswat_st.argtypes = [C.c_char_p, i64, C.c_char_p, i64, i64, i64, C.POINTER(C.c_double), C.c_void_p]
        L.oc_score2bit.restype = i64
        L.oc_score2bit.argtypes = [i64]
        L.oc_bit2e.restype = C.c_double
        L.oc_bit2e.argtypes = [i64, i64, i64, i64]
        L.oc_f2s.argtypes = [C.c_double, C.c_char_p, C.c_int]
        L.oc_fmt_idy.argtypes = [C.c_double, C.c_char_p, C.c_int]
        L.oc_index_build.restype = C.c_void_p
        L.oc_index_build.argtypes = [C.c_char_p, i64, C.c_char_p, C.c_char_p, i64, i64, i64, i64]
        L.oc_index_free.argtypes = [C.c_void_p]
        for f in ("oc_index_threshold", "oc_index_nlocus", "oc_index_nsoas"):
            getattr(L, f).restype = i64
            getattr(L, f).argtypes = [C.c_void_p]
        L.oc_index_set_threshold.argtypes = [C.c_void_p, i64]
        for f in ("oc_index_start", "oc_index_locus", "oc_index_soas"):
            getattr(L, f).restype = C.POINTER(C.c_uint32)
            getattr(L, f).argtypes = [C.c_void_p]
        L.oc_find_msav_m.restype = i64
        L.oc_find_msav_m.argtypes = [C.c_void_p, C.c_char_p, i64, C.c_void_p, i64, C.c_void_p, i64, p_i64, C.c_char_p]
        L.oc_blastp.restype = C.c_void_p
        L.oc_blastp.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_double, i64, C.c_double, i64, i64, i64,
                                i64, i64, i64, C.c_char_p, i64, i64, C.c_char_p, C.c_char_p]
        L.oc_result_free.argtypes = [C.c_void_p]
        L.oc_result_nrecs.restype = i64
        L.oc_result_nrecs.argtypes = [C.c_void_p]
        L.oc_result_recs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oc_result_ncands.restype = i64
        L.oc_result_ncands.argtypes = [C.c_void_p, i64]
        L.oc_result_nqueries.restype = i64
        L.oc_result_nqueries.argtypes = [C.c_void_p]
        L.oc_result_cands.argtypes = [C.c_void_p, i64, C.c_void_p]
        L.oc_result_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _b(s):
    return s if isinstance(s, bytes) else s.encode("latin-1")


def b62(a, b):
    return lib().oc_b62(ord(a) if isinstance(a, str) else a, ord(b) if isinstance(b, str) else b)


def b62_matrix():
    L = lib()
    return np.array([[L.oc_b62(i, j) for j in range(256)] for i in range(256)], dtype=np.int32)


def nr_tbl(gaa=AA9):
    out = (C.c_int * 512)()
    lib().oc_nr_tbl(_b(gaa), out)
    return list(out)


def spseeds(seq, ssd="111111", nr=AA9, mod=120000000, step=1):
    seq = _b(seq)
    cap = max(1, len(seq) * (ssd.count(",") + 1) * (nr.count("/") + 1))
    bk = np.zeros(cap, dtype=np.uint32)
    ps = np.zeros(cap, dtype=np.int32)
    n = lib().oc_spseeds(seq, len(seq), step, _b(nr), _b(ssd), mod, bk.ctypes.data, ps.ctypes.data)
    return [(int(bk[i]), int(ps[i])) for i in range(n)]


def seg(seq):
    seq = _b(seq)
    out = C.create_string_buffer(len(seq))
    lib().oc_seg(seq, len(seq), out)
    return out.raw[:len(seq)]


def qsort_perm(keys):
    k = np.asarray(keys, dtype=np.int64)
    perm = np.zeros(len(k), dtype=np.int32)
    lib().oc_qsort(k.ctypes.data, len(k), perm.ctypes.data)
    return perm.tolist()


def ungap(q, s, Qst, Sst, qlo=-1, slo=-1):
    q, s = _b(q), _b(s)
    out = np.zeros(6, dtype=np.int64)
    lib().oc_ungap(q, len(q), s, len(s), Qst, Sst, qlo, slo, out.ctypes.data)
    return tuple(int(x) for x in out)  # max_score, max_qst, max_qed, max_sst, max_sed, flag


def ungap_chain(q, s, locs):
    q, s = _b(q), _b(s)
    l = np.asarray(locs, dtype=np.int32).reshape(-1, 2)
    out = np.zeros(6, dtype=np.int64)
    lib().oc_ungap_chain(q, len(q), s, len(s), l.ctypes.data, len(l), out.ctypes.data)
    return tuple(int(x) for x in out)  # score, flag, x0, y0, x, y


def kswat_st(q, s, qst=0, sst=0, full=False):
    q, s = _b(q), _b(s)
    idy = C.c_double()
    out = np.zeros(10, dtype=np.int64)
    lib().oc_kswat_st(q, len(q), s, len(s), qst, sst, C.byref(idy), out.ctypes.data)
    r = (idy.value,) + tuple(int(x) for x in out[:8])
    return r + (int(out[8]), int(out[9])) if full else r


def score2bit(s):
    return int(lib().oc_score2bit(s))


def bit2e(D, li, lj, bit):
    return float(lib().oc_bit2e(D, li, lj, bit))


def f2s(e):
    buf = C.create_string_buffer(600)
    lib().oc_f2s(e, buf, 600)
    return buf.value.decode()


def fmt_idy(x):
    buf = C.create_string_buffer(600)
    lib().oc_fmt_idy(x, buf, 600)
    return buf.value.decode()


class Index:
    """One reference chunk indexed as Fasta.build_msav does (fsearch.py:2208-2280)."""

    def __init__(self, fasta_bytes, ssd="111111", nr=AA9, step=1, NC=120000000, start=0, end=-1):
        self.NC = NC
        fasta_bytes = _b(fasta_bytes)
        if end < 0:
            end = fasta_bytes.count(b"\n>") + 1
        self.h = lib().oc_index_build(fasta_bytes, len(fasta_bytes), _b(ssd), _b(nr), step, NC, start, end)

    def __del__(self):
        if getattr(self, "h", None):
            lib().oc_index_free(self.h)
            self.h = None

    @property
    def threshold(self):
        return int(lib().oc_index_threshold(self.h))

    @threshold.setter
    def threshold(self, t):
        lib().oc_index_set_threshold(self.h, t)

    def start(self):
        return np.ctypeslib.as_array(lib().oc_index_start(self.h), shape=(self.NC,)).copy()

    def locus(self):
        n = lib().oc_index_nlocus(self.h)
        if n == 0:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(lib().oc_index_locus(self.h), shape=(n,)).copy()

    def soas(self):
        n = lib().oc_index_nsoas(self.h)
        return np.ctypeslib.as_array(lib().oc_index_soas(self.h), shape=(n,)).copy()

    def find_msav_m(self, q, want_tuples=False, tcap=1 << 22):
        """-> candidates [[subject, score, qi, qj], ...] (+ seed-hit tuples, marks)."""
        q = _b(q)
        cap = 1 << 16
        cand = np.zeros((cap, 4), dtype=np.uint32)
        nt = C.c_int64(0)
        tuples = np.zeros((tcap if want_tuples else 1, 3), dtype=np.int32)
        marks = C.create_string_buffer(max(1, len(q)))
        n = lib().oc_find_msav_m(self.h, q, len(q), cand.ctypes.data, cap, tuples.ctypes.data if want_tuples else None,
                                 tcap if want_tuples else 0, C.byref(nt), marks)
        c = cand[:n].astype(np.int64).tolist()
        if want_tuples:
            assert nt.value <= tcap
            return c, tuples[:nt.value].copy(), np.frombuffer(marks.raw[:len(q)], dtype=np.int8).copy()
        return c


class Result:
    def __init__(self, h):
        self.h = h
        L = lib()
        n = L.oc_result_nrecs(h)
        ints = np.zeros((n, 13), dtype=np.int64)
        dbl = np.zeros((n, 2), dtype=np.float64)
        if n:
            L.oc_result_recs(h, ints.ctypes.data, dbl.ctypes.data)
        self.ints, self.dbl = ints, dbl
        s9 = np.zeros(9, dtype=np.int64)
        t3 = np.zeros(3, dtype=np.float64)
        L.oc_result_stats(h, s9.ctypes.data, t3.ctypes.data)
        keys = ["n_queries", "query_aa", "rows", "seed_hits", "groups", "ungap_steps", "cands", "alignments", "cells"]
        self.stats = dict(zip(keys, (int(x) for x in s9)))
        self.stats.update(t_index=float(t3[0]), t_seed=float(t3[1]), t_align=float(t3[2]))

    def cands(self, qrel):
        n = lib().oc_result_ncands(self.h, qrel)
        out = np.zeros((max(n, 1), 4), dtype=np.uint32)
        if n:
            lib().oc_result_cands(self.h, qrel, out.ctypes.data)
        return out[:n]

    @property
    def nqueries(self):
        return int(lib().oc_result_nqueries(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            lib().oc_result_free(self.h)
            self.h = None


class MergedResult:
    """The Results of consecutive query ranges behind Result's interface (blastp_parallel)."""

    def __init__(self, parts):
        self.parts = parts
        self.ints = np.vstack([p.ints for p in parts]) if parts else np.zeros((0, 13), dtype=np.int64)
        self.dbl = np.vstack([p.dbl for p in parts]) if parts else np.zeros((0, 2), dtype=np.float64)
        self.stats = {}
        for p in parts:
            for k, v in p.stats.items():
                self.stats[k] = self.stats.get(k, 0) + v
        self._first = np.cumsum([0] + [p.nqueries for p in parts])

    def cands(self, qrel):
        k = int(np.searchsorted(self._first, qrel, side="right")) - 1
        return self.parts[k].cands(qrel - int(self._first[k]))

    @property
    def nqueries(self):
        return int(self._first[-1])


def blastp_parallel(qry, ref, out_path="", threads=0, st=-1, ed=-1, min_piece=64, **kw):
    """blastp() with the query range cut into consecutive pieces, one thread each (ctypes releases the GIL; every piece indexes the
    reference itself, as every find_hit.py block does): same rows, records, candidate lists and counters as one call -- per-query
    results do not depend on the range (find_hit.py:107-146 relies on that).  Test infrastructure for the hosts with many cores."""
    import concurrent.futures
    import os
    data = open(qry, "rb").read()
    n = data.count(b"\n>") + (1 if data[:1] == b">" else 0)
    nref = n if ref == qry else None
    if nref is None:
        d2 = open(ref, "rb").read()
        nref = d2.count(b"\n>") + (1 if d2[:1] == b">" else 0)
    lo = min(max(0, st), n)
    hi = min(nref if ed < 0 else ed, n)
    threads = threads or min(32, os.cpu_count() or 1)
    k = max(1, min(threads, (hi - lo) // max(1, min_piece)))
    if k == 1 or kw.get("mode", "w") != "w":
        return blastp(qry, ref, out_path, st=st, ed=ed, **kw)
    lib().oc_b62(65, 65)   # (the score table is filled by its first user: not from several threads at once)
    cuts = [lo + (hi - lo) * i // k for i in range(k + 1)]
    outs = ["%s.part%d" % (out_path, i) if out_path else "" for i in range(k)]
    with concurrent.futures.ThreadPoolExecutor(max_workers=k) as ex:
        parts = list(ex.map(lambda i: blastp(qry, ref, outs[i], st=cuts[i], ed=cuts[i + 1], **kw), range(k)))
    if out_path:
        with open(out_path, "wb") as f:
            for o in outs:
                f.write(open(o, "rb").read())
                os.remove(o)
    return MergedResult(parts)


def blastp(qry, ref, out_path="", ssd="111111", nr=AA9, expect=1e-3, v=500, max_miss=1e-3, st=-1, ed=-1, rst=-1, red=-1,
           thr=-1, step=1, flt="T", ht=120000000, chk=50000, mode="w"):
    """End-to-end reference path for queries [st, ed) (fsearch.py blastp + entry_point)."""
    h = lib().oc_blastp(_b(qry), _b(ref), _b(ssd), _b(nr), expect, v, max_miss, st, ed, rst, red, thr, step, _b(flt), ht, chk,
                        _b(out_path), _b(mode))
    if not h:
        raise IOError("oracle blastp failed (cannot read %s / %s)" % (qry, ref))
    return Result(h)

"""find_orth -- counterpart of SwiftOrtho's bin/find_orth.py: from the hits of a find_hit search to the
orthology relations (`IP` in-paralogs, `OT` orthologs, `CO` co-orthologs with OrthoMCL-style normalised
scores); same flags, same stdout, line for line.

Design (nothing here is a transcription of the reference script, which streams text through three GNU
`sort` runs and mmap'd binary searches): the stage is COLUMNAR.  Its primary input is the array of
fixed-width hit records the search already holds (`fsearch.Hits.array()`, or the records gathered from the
GPUs) -- `relations_from_records()`; sequence ids become integer codes in byte order, so every "sort the
candidate file" of the reference is an integer lexsort, every per-query / per-taxon dictionary a segmented
numpy reduction, and the co-ortholog cross products are index arithmetic.  A 16- or 12-column text file
(the CLI's `-i`) is tokenised with numpy into the same columns -- `find_orth()` -- and takes the same path.

Behaviour pinned by the goldens (stdout of the REAL bin/find_orth.py, tools/refharness/make_orth_goldens.py);
what the reference does, with its line numbers:
  * rows whose numeric columns do not parse are skipped; filters: query coverage (1 + |qed - qst|) / qlen
    >= -c and identity >= -y; 12-column input takes the query length from the first row of that query id
    (156-234);  -n no | bsr (bit score / first kept score of the query id) | bal (bit score / alignment length);
  * rows are grouped by CONSECUTIVE query id; per (group, subject) the largest score, at the place of the
    subject's first row;
  * per group: best score per subject taxon and best out-of-taxon score; a same-taxon hit at least as good
    as the latter, not the query itself, is an in-paralog candidate (both orientations); an other-taxon hit
    equal to its taxon's best an ortholog candidate, else a co-ortholog candidate (298-348);
  * a candidate pair is a relation iff proposed exactly TWICE; its score is the mean of the two -- the LAST
    pair of the byte-sorted candidate file gets the maximum instead (351-377);
  * in-paralog normaliser per taxon: mean over the pairs with an ortholog on either side, else over all
    (507-543); co-orthologs: for every ortholog pair every (in-paralog-or-self) x (in-paralog-or-self)
    combination that is a co-ortholog candidate in that orientation, the pair itself included, provided
    either side has an in-paralog (547-617);
  * OT / CO lines are normalised inside runs of one query taxon by the mean score per subject taxon;
    repeats inside a run are dropped, except that one repeat of the run's first pair survives (681-762).
Scores print as Python's str(float), like the reference under Python 3.
"""
import os
import sys

import numpy as np

DEFAULTS = {'-i': '', '-c': .5, '-y': 0, '-n': 'no', '-t': 'n', '-a': '4', '-T': './tmp/', '-s': '|'}


def manual_print(prog='find_orth.py'):
    print('Usage:')
    print('    python %s -i foo.sc [-c .5] [-y 50] [-n no]' % prog)
    print('Parameters:')
    print('  -i: tab-delimited file which contain 14 columns')
    print('  -c: min coverage of sequence [0~1]')
    print('  -y: identity [0~100]')
    print('  -n: normalization score [no|bsr|bal]. bsr: bit sore ratio; bal:  bit score over anchored length. Default: no')
    print('  -a: cpu number for sorting. Default: 1 (accepted, unused: nothing is sorted on disk)')
    print('  -t: keep tmpdir[y|n]. Default: n (accepted, unused)')
    print('  -T: tmpdir for sort command. Default: ./tmp/ (accepted, unused)')
    print('  -s: separator between taxa and sequence id. Default is |.')


# ---------------------------------------------------------------------------------------------------------
# columns
# ---------------------------------------------------------------------------------------------------------
class HitColumns:
    """The columns this stage reads, one numpy array each, rows in file order.  `names` holds every distinct sequence
    id (bytes, ascending byte order == the order of `LC_ALL=C sort` and of Python's string compare) and `q` / `s` index it."""

    def __init__(self, names, q, s, idy, aln, qst, qed, score, qlen):
        self.names, self.q, self.s = names, q, s
        self.idy, self.aln, self.qst, self.qed, self.score, self.qlen = idy, aln, qst, qed, score, qlen


def _codes(qnames, snames):
    """two id arrays (bytes) -> (sorted distinct ids, codes of the first, codes of the second)"""
    w = max(qnames.dtype.itemsize, snames.dtype.itemsize, 1)
    both = np.concatenate([qnames.astype('S%d' % w), snames.astype('S%d' % w)])
    names, inv = np.unique(both, return_inverse=True)
    inv = inv.astype(np.int64)
    return names, inv[:len(qnames)], inv[len(qnames):]


def columns_from_records(hits, query_ids, subject_ids):
    """`hits`: structured array of so_hit records (fsearch.HIT_DTYPE: qidx, sidx, identity, aln, qst, qed, bit, qlen ...);
    `query_ids` / `subject_ids`: the id (header up to the first blank) of every sequence of the two FASTA files.
    The identity column is the 2-decimal value the text row would carry (`%f` cut after two decimals)."""
    qn = np.asarray(query_ids, dtype=np.bytes_)
    sn = np.asarray(subject_ids, dtype=np.bytes_)
    names, qmap, smap = _codes(qn, sn)
    idy = np.asarray(hits['identity'], dtype=np.float64)
    idy = np.floor(np.round(idy, 6) * 100. + 1e-7) / 100.
    f = lambda k: np.asarray(hits[k], dtype=np.float64)
    return HitColumns(names, qmap[np.asarray(hits['qidx'], dtype=np.int64)], smap[np.asarray(hits['sidx'], dtype=np.int64)], idy, f('aln'),
                      f('qst'), f('qed'), f('bit'), f('qlen'))


def fasta_ids(data):
    """ids of a FASTA file the way the search names its rows (fsearch.py:1543-1553, 3236-3240): a record starts at a '>'
    that follows a newline (or opens the file); the id is its header line up to the first blank."""
    b = np.frombuffer(data, dtype=np.uint8)
    if b.size == 0:
        return []
    st = np.flatnonzero((b[1:] == 62) & (b[:-1] == 10)) + 1
    st = np.concatenate([[0], st])
    out = []
    for p in st.tolist():
        e = data.find(b'\n', p)
        h = data[p + 1:e if e >= 0 else len(data)]
        out.append(h.split(b' ')[0])
    return out


def _to_float(fields):
    """fixed-width bytes array -> (float64 values, parsed-ok mask) with Python's float() grammar"""
    try:
        return fields.astype(np.float64), np.ones(len(fields), dtype=bool)
    except ValueError:
        pass
    vals = np.zeros(len(fields), dtype=np.float64)
    ok = np.ones(len(fields), dtype=bool)
    for i, t in enumerate(fields.tolist()):
        try:
            vals[i] = float(t.decode('latin-1'))
        except ValueError:
            ok[i] = False
    return vals, ok


_FLOAT_COLS = (('idy', 2), ('aln', 3), ('mis', 4), ('gop', 5), ('qst', 6), ('qed', 7), ('sst', 8), ('sed', 9), ('evalue', 10), ('score', 11))


def _columns_native(data):
    """The same tokenisation by libsohit's threaded scanner (include/sohit.h so_tsv_*, csrc/tsv.hip): field bounds, plain decimal
    numbers through strtod (correctly rounded, like float()), id codes through a hash map.  Returns None -- and the numpy path below
    runs -- when the library is not built, the input is tiny, or ANY numeric field is not a plain decimal number (Python's float()
    grammar is wider: 'inf', '1_0', ...; real files never hold such fields)."""
    n = data.count(b'\n')
    if n < int(os.environ.get('SOHIT_TSV_MIN', '4096')) or os.environ.get('SOHIT_TSV_NATIVE', '1') == '0':   # (tests force / forbid the native path)
        return None
    try:
        from . import _lib
        L = _lib.load()
    except Exception:
        return None
    import ctypes as C
    ptr = lambda a: C.c_void_p(a.ctypes.data)
    buf = np.frombuffer(data, dtype=np.uint8)
    ls = np.empty(n + 1, dtype=np.int64)
    if L.so_tsv_lines(ptr(buf), len(data), ptr(ls), n + 1) != n:
        return None
    ncol = 14
    cols = np.arange(ncol, dtype=np.int32)
    # ids: field bounds; idy, aln, qst, qed, score, qlen: value + status (3); the other numeric columns only decide whether the row
    # parses: status (2).  The [14][n] arrays below are np.empty: what the scanner is not asked for is never touched (470 -> 130 MB).
    numeric = np.array([0, 0, 3, 3, 2, 2, 3, 3, 2, 2, 2, 3, 3, 2], dtype=np.uint8)
    ntab = np.empty(n, dtype=np.int32)
    beg = np.empty((ncol, n), dtype=np.int64)
    ln = np.empty((ncol, n), dtype=np.int32)
    val = np.empty((ncol, n), dtype=np.float64)
    st = np.empty((ncol, n), dtype=np.uint8)
    if L.so_tsv_scan(ptr(buf), len(data), ptr(ls), n, ncol, ptr(cols), ptr(numeric), ptr(ntab), ptr(beg), ptr(ln), ptr(val), ptr(st)) != 0:
        return None
    if np.any(ntab < 1):
        raise ValueError('find_orth: a row has fewer than two columns')
    full = ntab >= 11
    wide = ntab >= 13
    if np.any(st[2:12][:, full] == 2) or np.any(st[12:14][:, wide] == 2):
        return None                                # a field only Python's float() can judge
    ok = full.copy()
    vals = {}
    for name, k in _FLOAT_COLS:
        vals[name] = val[k]
        ok &= st[k] == 0
    qlen = np.where(wide, val[12], 0.)
    ok &= ~wide | ((st[12] == 0) & (st[13] == 0))
    qc = np.empty(n, dtype=np.int64)
    sc = np.empty(n, dtype=np.int64)
    nb = np.empty(2 * n, dtype=np.int64)
    nlen = np.empty(2 * n, dtype=np.int32)
    nd = L.so_tsv_codes(ptr(buf), n, ptr(beg[0]), ptr(ln[0]), ptr(beg[1]), ptr(ln[1]), ptr(qc), ptr(sc), ptr(nb), ptr(nlen), 2 * n)
    if nd <= 0:
        return None
    w = max(int(nlen[:nd].max()), 1)
    names = np.array([data[b:b + l] for b, l in zip(nb[:nd].tolist(), nlen[:nd].tolist())], dtype='S%d' % w)
    return _finish_columns(names, qc, sc, vals, ok, wide, qlen)


def columns_from_text(data):
    """tab-separated rows (bytes of a 12-column blast -m8 or 16-column find_hit file) -> HitColumns.  Tokenised with numpy:
    one pass finds the line ends and tabs, each needed field is gathered into a fixed-width array and converted at once."""
    data = data.replace(b'\r\n', b'\n').replace(b'\r', b'\n')   # the reference reads in text mode (universal newlines)
    if data and not data.endswith(b'\n'):
        data = data[:-1] + b'\n'                                # ... and cuts the last character of every line, newline or not
    fast = _columns_native(data)
    if fast is not None:
        return fast
    buf = np.frombuffer(data, dtype=np.uint8)
    nl = np.flatnonzero(buf == 10)
    n = len(nl)
    empty = np.zeros(0, dtype=np.float64)
    if n == 0:
        return HitColumns(np.zeros(0, dtype='S1'), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), empty, empty, empty, empty, empty, empty)
    ls = np.concatenate([[0], nl[:-1] + 1])        # line starts
    tabs = np.flatnonzero(buf == 9)
    t0 = np.searchsorted(tabs, ls)                 # first tab of every line
    t1 = np.searchsorted(tabs, nl)                 # one past its last tab
    ntab = t1 - t0
    if np.any(ntab < 1):
        raise ValueError('find_orth: a row has fewer than two columns')
    tabs_pad = np.concatenate([tabs, np.full(16, len(buf), dtype=tabs.dtype)])

    def field(k):   # (start, end) of column k; rows with too few columns get an empty field
        st = ls if k == 0 else tabs_pad[t0 + k - 1] + 1
        en = np.where(ntab > k, tabs_pad[t0 + k], nl)
        have = ntab >= k
        return np.where(have, st, 0), np.where(have, np.maximum(en, st), 0)

    def gather(k):
        st, en = field(k)
        ln = en - st
        w = int(ln.max()) if n else 1
        w = max(w, 1)
        if n * w > (1 << 31):
            raise MemoryError('find_orth: a column is too wide to tokenise in memory')
        idx = st[:, None] + np.arange(w)[None, :]
        m = np.arange(w)[None, :] < ln[:, None]
        arr = np.where(m, buf[np.minimum(idx, len(buf) - 1)], 0).astype(np.uint8)
        return np.ascontiguousarray(arr).view('S%d' % w).ravel()

    qn, sn = gather(0), gather(1)
    ok = ntab >= 11                                # `map(float, j[2:12])` needs columns 3..12
    vals = {}
    for name, k in (('idy', 2), ('aln', 3), ('mis', 4), ('gop', 5), ('qst', 6), ('qed', 7), ('sst', 8), ('sed', 9), ('evalue', 10), ('score', 11)):
        v, good = _to_float(np.char.strip(gather(k)))
        vals[name] = v
        ok &= good
    wide = ntab >= 13                              # len(j) > 13: columns 13 and 14 are the sequence lengths
    qlen = np.zeros(n, dtype=np.float64)
    if wide.any():
        v, good = _to_float(np.char.strip(gather(12)))
        v2, good2 = _to_float(np.char.strip(gather(13)))
        qlen = np.where(wide, v, 0.)
        ok &= ~wide | (good & good2)
    names, q, s = _codes(qn, sn)
    return _finish_columns(names, q, s, vals, ok, wide, qlen)


def _finish_columns(names, q, s, vals, ok, wide, qlen):
    """rows that parsed (`ok`) -> HitColumns; 12-column rows take their query length from the first such row of the query"""
    used = ('idy', 'aln', 'qst', 'qed', 'score')   # (the other numeric columns only decide whether a row parses)
    if ok.all():                                   # the usual file: nothing to drop, no copies
        col = {k: vals[k] for k in used}
    else:
        keep = np.flatnonzero(ok)
        q, s = q[keep], s[keep]
        col = {k: vals[k][keep] for k in used}
        qlen, wide = qlen[keep], wide[keep]
    if not wide.all():
        # 12-column rows: the length of a query is max(qst, qed) of the first such row that carries its id
        narrow = np.flatnonzero(~wide)
        uq, first = np.unique(q[narrow], return_index=True)
        est = np.zeros(len(names), dtype=np.float64)
        fr = narrow[first]
        est[uq] = np.maximum(col['qst'][fr], col['qed'][fr])
        qlen = np.where(wide, qlen, est[q])
    return HitColumns(names, q, s, col['idy'], col['aln'], col['qst'], col['qed'], col['score'], qlen)


# ---------------------------------------------------------------------------------------------------------
# segmented helpers
# ---------------------------------------------------------------------------------------------------------
def _group_max(key, val, floor=None):
    """-> (inverse, per-key maximum of val); `floor`: every maximum starts from it"""
    uk, inv = np.unique(key, return_inverse=True)
    order = np.argsort(inv, kind='stable')
    bounds = np.flatnonzero(np.diff(inv[order], prepend=-1))
    mx = np.maximum.reduceat(val[order], bounds) if len(order) else np.zeros(0)
    if floor is not None:
        mx = np.maximum(mx, floor)
    return inv, mx


def _pairs_proposed_twice(a, b, sco):
    """candidate triples -> the pairs that occur exactly twice, in (a, b) order, scored with the mean of the two proposals;
    when the very last pair of that order is one of them it takes the larger proposal instead"""
    if len(a) == 0:
        return a, b, sco
    order = np.lexsort((b, a))
    a, b, sco = a[order], b[order], sco[order]
    head = np.flatnonzero(np.concatenate([[True], (a[1:] != a[:-1]) | (b[1:] != b[:-1])]))
    size = np.diff(np.concatenate([head, [len(a)]]))
    two = head[size == 2]
    s0, s1 = sco[two], sco[two + 1]
    val = ((0. + s0) + s1) / 2.
    if len(two) and two[-1] + 2 == len(a):
        val[-1] = max(s0[-1], s1[-1])
    return a[two], b[two], val


def _taxa(names, sep):
    """taxon code of every name (the part before the first separator) + the distinct taxa"""
    if len(names) == 0:
        return np.zeros(0, dtype=np.int64), names
    sepb = sep.encode('utf-8')
    assert bool(np.all(np.char.find(names, sepb) >= 0)), 'an id lacks the taxon separator %r' % sep
    head = np.char.partition(names, sepb)[:, 0]
    tx, inv = np.unique(head, return_inverse=True)
    return inv.astype(np.int64), tx


# ---------------------------------------------------------------------------------------------------------
# the stage
# ---------------------------------------------------------------------------------------------------------
def relations(cols, coverage=.5, identity=0., norm='no', sep='|'):
    """HitColumns -> output lines (bytes, no newline) in the reference's order: IP, then OT, then CO"""
    names = cols.names
    M = max(len(names), 1)
    tax, taxa = _taxa(names, sep)
    T = max(len(taxa), 1)
    # ---- row filter and score ------------------------------------------------------------------------------
    with np.errstate(divide='ignore', invalid='ignore'):
        qcv = (1. + np.abs(cols.qed - cols.qst)) / cols.qlen
    keep = np.flatnonzero(~((qcv < coverage) | (cols.idy < identity)))
    q, s, bit, aln = cols.q[keep], cols.s[keep], cols.score[keep], cols.aln[keep]
    if norm == 'bsr':
        uq, first = np.unique(q, return_index=True)
        ref = np.zeros(M, dtype=np.float64)
        ref[uq] = bit[first]
        sco = bit / ref[q]
    elif norm == 'bal':
        sco = bit / aln
    else:
        sco = bit
    # ---- groups = runs of one query id; per (group, subject) the best score at the subject's first row -------------
    n = len(q)
    if n == 0:
        return []
    run = np.cumsum(np.concatenate([[0], q[1:] != q[:-1]]).astype(np.int64))
    key = run * M + s
    uk, first, inv = np.unique(key, return_index=True, return_inverse=True)
    best = np.full(len(uk), -np.inf)
    np.maximum.at(best, inv, sco)
    o = np.argsort(first, kind='stable')
    rows = first[o]
    g_run, g_q, g_s, g_sco = run[rows], q[rows], s[rows], best[o]
    # ---- candidates ------------------------------------------------------------------------------------------
    qtx, stx = tax[g_q], tax[g_s]
    same = qtx == stx
    inv_t, tmax = _group_max(g_run * T + stx, g_sco, floor=0.)
    nrun = int(g_run[-1]) + 1
    out_max = np.zeros(nrun, dtype=np.float64)
    np.maximum.at(out_max, g_run[~same], g_sco[~same])
    a, b = np.minimum(g_q, g_s), np.maximum(g_q, g_s)
    is_ip = same & (g_sco >= out_max[g_run]) & (g_q != g_s)
    is_ot = ~same & (g_sco >= tmax[inv_t])
    is_co = ~same & ~is_ot
    ip_a = np.concatenate([a[is_ip], b[is_ip]])
    ip_b = np.concatenate([b[is_ip], a[is_ip]])
    ip_s = np.concatenate([g_sco[is_ip], g_sco[is_ip]])
    ot_a, ot_b, ot_s = _pairs_proposed_twice(a[is_ot], b[is_ot], g_sco[is_ot])
    ip_a, ip_b, ip_s = _pairs_proposed_twice(ip_a, ip_b, ip_s)
    # ---- in-paralog normalisers ---------------------------------------------------------------------------------
    has_ot = np.zeros(M, dtype=bool)
    has_ot[ot_a] = True
    has_ot[ot_b] = True
    fwd = ip_a < ip_b
    ta = tax[ip_a[fwd]]
    all_sum = np.bincount(ta, weights=ip_s[fwd], minlength=T)
    all_cnt = np.bincount(ta, minlength=T)
    near = has_ot[ip_a[fwd]] | has_ot[ip_b[fwd]]
    near_sum = np.bincount(ta[near], weights=ip_s[fwd][near], minlength=T)
    near_cnt = np.bincount(ta[near], minlength=T)
    with np.errstate(divide='ignore', invalid='ignore'):
        ip_avg = np.where(near_cnt > 0, near_sum / np.maximum(near_cnt, 1), all_sum / np.maximum(all_cnt, 1))
    # ---- co-orthologs -------------------------------------------------------------------------------------------
    co_a = np.zeros(0, dtype=np.int64)
    co_b, co_s = co_a, np.zeros(0, dtype=np.float64)
    if len(ip_a) and is_co.any() and len(ot_a):
        ck = a[is_co] * M + b[is_co]
        cu, cinv = np.unique(ck, return_inverse=True)
        cbest = np.full(len(cu), -np.inf)
        np.maximum.at(cbest, cinv, g_sco[is_co])
        lo_q, hi_q = np.searchsorted(ip_a, ot_a, 'left'), np.searchsorted(ip_a, ot_a, 'right')
        lo_s, hi_s = np.searchsorted(ip_a, ot_b, 'left'), np.searchsorted(ip_a, ot_b, 'right')
        nq, ns = hi_q - lo_q, hi_s - lo_s
        use = np.flatnonzero((nq > 0) | (ns > 0))
        cnt = (nq[use] + 1) * (ns[use] + 1)
        tot = int(cnt.sum())
        if tot:
            pid = np.repeat(np.arange(len(use)), cnt)
            local = np.arange(tot) - np.repeat(np.cumsum(cnt) - cnt, cnt)
            u = use[pid]
            qi, si = local // (ns[u] + 1), local % (ns[u] + 1)
            qip = np.where(qi < nq[u], ip_b[np.minimum(lo_q[u] + qi, len(ip_b) - 1)], ot_a[u])
            sip = np.where(si < ns[u], ip_b[np.minimum(lo_s[u] + si, len(ip_b) - 1)], ot_b[u])
            k = qip * M + sip
            pos = np.minimum(np.searchsorted(cu, k), len(cu) - 1)
            hit = cu[pos] == k
            co_a, co_b, co_s = qip[hit], sip[hit], cbest[pos[hit]]
    # ---- text ----------------------------------------------------------------------------------------------------
    lines = []
    nm = names.tolist()
    fmt = _PairFormatter(nm)
    avg = ip_avg[tax[ip_a[fwd]]]
    with np.errstate(divide='ignore', invalid='ignore'):
        val = ip_s[fwd] / avg
    ok = avg != 0    # the reference's division raises there and the line is skipped
    lines.extend(fmt.lines(b'IP', ip_a[fwd][ok], ip_b[fwd][ok], val[ok]))
    for kind, (pa, pb, ps) in ((b'OT', (ot_a, ot_b, ot_s)), (b'CO', (co_a, co_b, co_s))):
        if len(pa) == 0:
            continue
        blk = np.cumsum(np.concatenate([[0], tax[pa][1:] != tax[pa][:-1]]).astype(np.int64))
        # occurrence number of every pair inside its block
        pk = pa * M + pb
        order = np.lexsort((np.arange(len(pa)), pk, blk))
        sk = (blk[order], pk[order])
        newgrp = np.concatenate([[True], (sk[0][1:] != sk[0][:-1]) | (sk[1][1:] != sk[1][:-1])])
        start = np.maximum.accumulate(np.where(newgrp, np.arange(len(pa)), 0))
        occ = np.empty(len(pa), dtype=np.int64)
        occ[order] = np.arange(len(pa)) - start
        bfirst = np.flatnonzero(np.concatenate([[True], blk[1:] != blk[:-1]]))
        is_first_pair = pk == pk[bfirst][blk]
        keepr = np.flatnonzero((occ == 0) | (is_first_pair & (occ == 1)))
        pa, pb, ps, blk = pa[keepr], pb[keepr], ps[keepr], blk[keepr]
        gk = blk * T + tax[pb]
        gu, ginv = np.unique(gk, return_inverse=True)
        gsum = np.bincount(ginv, weights=ps, minlength=len(gu))
        gcnt = np.bincount(ginv, minlength=len(gu)).astype(np.float64)
        val = ps / (gsum / gcnt)[ginv]
        lines.extend(fmt.lines(kind, pa, pb, val))
    return lines


class _PairFormatter:
    """lines  kind \\t id \\t id \\t repr(score)  for arrays of id codes and scores: through libsohit (so_format_pairs: Python's repr of a
    float reproduced digit for digit, rows written by threads) when it is there and the section is large, else one by one"""

    def __init__(self, names):
        self.nm = names
        self.blob = None

    def lines(self, kind, x, y, v):
        n = len(x)
        if n == 0:
            return []
        if n >= int(os.environ.get('SOHIT_TSV_MIN', '4096')) and os.environ.get('SOHIT_TSV_NATIVE', '1') != '0':
            try:
                from . import _lib
                L = _lib.load()
                import ctypes as C
                if self.blob is None:
                    self.off = np.zeros(len(self.nm) + 1, dtype=np.int64)
                    np.cumsum([len(t) for t in self.nm], out=self.off[1:])
                    self.blob = b''.join(self.nm)
                xa, ya = np.ascontiguousarray(x, dtype=np.int64), np.ascontiguousarray(y, dtype=np.int64)
                va = np.ascontiguousarray(v, dtype=np.float64)
                ptr = lambda a: C.c_void_p(a.ctypes.data)
                cap = int(n * (len(kind) + 32) + (self.off[xa + 1] - self.off[xa]).sum() + (self.off[ya + 1] - self.off[ya]).sum())
                out = np.empty(cap, dtype=np.uint8)
                w = L.so_format_pairs(kind, len(kind), self.blob, ptr(self.off), ptr(xa), ptr(ya), ptr(va), n, ptr(out), cap)
                if w > 0:
                    return out[:w - 1].tobytes().split(b'\n')
            except Exception:
                pass
        nm = self.nm
        return [kind + b'\t' + nm[a] + b'\t' + nm[b] + b'\t' + repr(c).encode() for a, b, c in zip(x.tolist(), y.tolist(), v.tolist())]


def relations_from_records(hits, query_ids, subject_ids, coverage=.5, identity=0., norm='no', sep='|'):
    """the stage on the search's own hit records (no text round trip) -> output lines (bytes)"""
    return relations(columns_from_records(hits, query_ids, subject_ids), coverage, identity, norm, sep)


def find_orth(src, coverage=.5, identity=0., norm='no', sep='|'):
    """`src`: an open text/binary file, bytes, or an iterable of lines -> output lines as str ('IP|OT|CO\\tqid\\tsid\\tscore')"""
    if hasattr(src, 'read'):
        data = src.buffer.read() if hasattr(src, 'buffer') else src.read()
    elif isinstance(src, (bytes, bytearray)):
        data = bytes(src)
    else:
        data = ''.join(src)
    if isinstance(data, str):
        data = data.encode('utf-8')
    return [l.decode('utf-8') for l in relations(columns_from_text(data), coverage, identity, norm, sep)]


def parse(argv):
    from .fsearch import parse_flags
    return parse_flags(argv, DEFAULTS)


def main(argv=None):
    argv = list(sys.argv if argv is None else argv)
    args = parse(argv)
    if args['-i'] == '':
        manual_print(argv[0] if argv else 'find_orth.py')
        raise SystemExit()
    try:
        qry, coverage, identity, norm, sep = args['-i'], float(args['-c']), float(args['-y']), args['-n'], args['-s']
        int(args['-a'])
    except Exception:
        manual_print(argv[0] if argv else 'find_orth.py')
        raise SystemExit()
    with open(qry, 'rb') as f:
        lines = relations(columns_from_text(f.read()), coverage, identity, norm, sep)
    out = sys.stdout.buffer
    if lines:
        out.write(b'\n'.join(lines) + b'\n')
    out.flush()
    return 0


if __name__ == '__main__':
    if __package__ in (None, ''):
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from swiftortho_amd.find_orth import main as _m
        sys.exit(_m())
    sys.exit(main())

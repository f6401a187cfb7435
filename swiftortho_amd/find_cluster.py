"""find_cluster -- counterpart of SwiftOrtho's bin/find_cluster.py for `-a mcl`: from the orthology
relations file (find_orth output) to ortholog groups, one tab-separated group per line, same flags,
same stdout.

Last stage of BASELINE config 5 (find_hit -> find_orth -> find_cluster -a mcl -I 1.5); a SURVEY.md 8f
"next" row.  The Markov-cluster iteration -- the one data-parallel part of this stage: column normalisation,
sparse x sparse expansion, inflation, pruning on float32 CSR matrices -- runs on the GPU (libsohit `so_mcl`,
csrc/mcl.hip) with scipy's arithmetic order, because the pruning decisions, and therefore the groups, depend
on it; the graph bookkeeping around it (best-neighbour components, batching, the read-out) is host code.
There is no CPU path for the loop: without the HIP library `-a mcl` fails.  Only `mcl` is provided; the
reference's default `-a apc` and `-a sap` (affinity propagation) are refused with a message and exit code 2.

Reference behaviour reproduced (bin/find_cluster.py, `cnc` 1470-1673, `mcl_xyz` 1425-1467, `mcl`
652-689, `normalize` 636-646), including its accidents, because they decide which genes appear:
  * rows `TYPE a b score` (or `a b score`); only rows with a <= b are used;
  * level 1: every gene is linked to its best-scoring neighbour(s) (ties all kept); connected
    components of that graph are numbered in discovery order;
  * level 2: components joined by any edge are merged -- but an edge is only seen when BOTH of its
    level-1 component numbers are non-zero (`if X and Y`), so component 0 never merges;
  * an edge is kept when both ends carry the same level-2 number and that number is non-zero
    (`if cx and cy and cx == cy`): everything in level-2 group 0 is dropped from the output, and all
    genes outside any level-2 group share the number -1 and are clustered together as one block;
  * the kept edges, sorted like `LC_ALL=C sort -n`, go through MCL in batches cut at group changes
    once more than 10^7 edges have accumulated;
  * MCL on a (D+1)x(D+1) float32 matrix with self-loops = the largest incident weight: column
    normalisation, squaring, inflation, pruning below 1e-5, at most 100 rounds, convergence test every
    5th round against the normalised matrix of that round; groups = connected components of the
    entries > 1e-5 -- read, as the reference does, by zipping the coordinates of the non-zero entries
    with the raw data array (which still holds the pruned zeros when the round limit was hit).
"""
import sys

import numpy as np

DEFAULTS = {'-i': '', '-d': '0.5', '-p': '-10000', '-I': '1.5', '-a': 'apc', '-t': '2', '-b': '25000000'}


def manual_print(prog='find_cluster.py'):
    print('Usage:')
    print('    python %s -i foo.xyz -d 0.5' % prog)
    print('Parameters:')
    print('  -i: tab-delimited file which contain 3 columns')
    print('  -d: damp (affinity propagation; not provided here)')
    print('  -p: parameter of preference for apc (not provided here)')
    print('  -I: inflation parameter for mcl')
    print('  -a: algorithm (mcl)')
    print('  -t: cpu number')


class _Graph:
    """undirected graph with networkx's iteration order: nodes in insertion order, components in order of their first node"""

    def __init__(self):
        self.adj = {}

    def add_edge(self, u, v):
        if u not in self.adj:
            self.adj[u] = {}
        if v not in self.adj:
            self.adj[v] = {}
        self.adj[u][v] = 1
        self.adj[v][u] = 1

    @classmethod
    def from_pairs(cls, pairs):
        """the graph after `add_edge(u, v)` for every (u, v) of `pairs` in order, built column-wise: a node's place is its first
        appearance in the sequence u0, v0, u1, v1, ...; its neighbour dict holds the other ends of its edges in that same order
        (a repeated edge keeps its first place, as a repeated assignment does)"""
        g = cls()
        if isinstance(pairs, tuple) and len(pairs) == 2 and isinstance(pairs[0], np.ndarray):
            e = np.stack(pairs, axis=1)        # (u column, v column)
        else:
            e = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
        if not len(e):
            return g
        src = e.reshape(-1)                    # u0, v0, u1, v1, ...
        dst = e[:, ::-1].reshape(-1)           # v0, u0, v1, u1, ...
        order = np.argsort(src, kind='stable')
        s_sorted = src[order]
        starts = np.flatnonzero(np.r_[True, s_sorted[1:] != s_sorted[:-1]])
        ends = np.r_[starts[1:], len(src)]
        first = order[starts]                  # a node's first position in the sequence
        d_sorted = dst[order].tolist()
        nodes = s_sorted[starts].tolist()
        for k in np.argsort(first, kind='stable').tolist():
            g.adj[nodes[k]] = dict.fromkeys(d_sorted[starts[k]:ends[k]], 1)
        return g

    def components(self):
        seen = set()
        for s in self.adj:
            if s in seen:
                continue
            comp, frontier = {s}, [s]
            seen.add(s)
            while frontier:
                nxt = []
                for u in frontier:
                    for v in self.adj[u]:
                        if v not in seen:
                            seen.add(v)
                            comp.add(v)
                            nxt.append(v)
                frontier = nxt
            yield comp


def _rows(lines):
    for i in lines:
        j = i[:-1].split('\t')
        if len(j) == 4:
            x, y, z = j[1:4]
        else:
            x, y, z = j[:3]
        if x > y:
            continue
        yield x, y, z


def block_matrix(edge_lines):
    """The matrix mcl_xyz (find_cluster.py:1425-1467) hands to the Markov loop, built columnar: genes numbered by first appearance over
    ALL lines of the batch, rows with x <= y only, (x, y) and (y, x) carry the weight of the pair's LAST line, the diagonal the largest
    incident weight, weights in float32, explicit zeros not stored; (D + 1) x (D + 1) with an empty last row, canonical CSR (columns
    ascending).  -> (gene names by number, indptr int64, indices int32, data float32)"""
    cols = [l.split('\t', 3)[:3] for l in edge_lines]
    xs = np.array([c[0] for c in cols], dtype=object)
    ys = np.array([c[1] for c in cols], dtype=object)
    inter = np.empty(2 * len(cols), dtype=object)
    inter[0::2], inter[1::2] = xs, ys
    names = list(dict.fromkeys(inter.tolist()))            # numbering in order of first appearance
    number = {g: k for k, g in enumerate(names)}
    dmx = len(names) + 1
    use = np.array([c[0] <= c[1] for c in cols], dtype=bool)
    X = np.fromiter((number[g] for g in xs[use].tolist()), dtype=np.int64, count=int(use.sum()))
    Y = np.fromiter((number[g] for g in ys[use].tolist()), dtype=np.int64, count=int(use.sum()))
    Z = np.array([float(c[2]) for c, u in zip(cols, use.tolist()) if u], dtype=np.float64).astype(np.float32)
    return names, *_matrix_from_numbers(X, Y, Z, dmx)


def block_matrix_arrays(gx, gy, z, gene_names):
    """block_matrix for rows that are already columns: gx / gy = gene numbers (any integer labelling) of the batch's rows in batch order
    (every row has x <= y), z = their weights (float64); genes are renumbered by first appearance over the batch (x before y)."""
    seq = np.empty(2 * len(gx), dtype=np.int64)
    seq[0::2], seq[1::2] = gx, gy
    vals, first = np.unique(seq, return_index=True)
    order = np.argsort(first, kind='stable')
    rank = np.empty(len(vals), dtype=np.int64)
    rank[order] = np.arange(len(vals))
    X, Y = rank[np.searchsorted(vals, gx)], rank[np.searchsorted(vals, gy)]
    names = [gene_names[g] for g in vals[order].tolist()]
    return names, *_matrix_from_numbers(X, Y, np.asarray(z, dtype=np.float64).astype(np.float32), len(names) + 1)


def _matrix_from_numbers(X, Y, Z, dmx):
    if np.any(X == Y):
        return _block_matrix_sequential(X, Y, Z, dmx)   # self loops overwrite the diagonal in line order: rare, done one by one
    # off-diagonal: the last line of a pair wins
    key = X * dmx + Y
    last = len(key) - 1 - np.unique(key[::-1], return_index=True)[1]
    px, py, pz = X[last], Y[last], Z[last]
    # diagonal: the largest weight seen at either end (a zero start: weights <= 0 never set it)
    diag = np.zeros(dmx, dtype=np.float32)
    np.maximum.at(diag, X, Z)
    np.maximum.at(diag, Y, Z)
    dn = np.flatnonzero(diag > 0)
    r = np.concatenate([px, py, dn])
    c = np.concatenate([py, px, dn])
    v = np.concatenate([pz, pz, diag[dn]])
    keep = v != 0
    r, c, v = r[keep], c[keep], v[keep]
    o = np.lexsort((c, r))
    r, c, v = r[o], c[o], v[o]
    indptr = np.zeros(dmx + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=dmx), out=indptr[1:])
    return indptr, c.astype(np.int32), v.astype(np.float32)


def _block_matrix_sequential(X, Y, Z, dmx):
    cell = {}
    zero = np.float32(0)
    for a, b, z in zip(X.tolist(), Y.tolist(), Z):
        cell[(a, b)] = z
        cell[(b, a)] = z
        if cell.get((a, a), zero) < z:
            cell[(a, a)] = z
        if cell.get((b, b), zero) < z:
            cell[(b, b)] = z
    keys = sorted(k for k, v in cell.items() if v != 0)
    r = np.array([k[0] for k in keys], dtype=np.int64)
    indptr = np.zeros(dmx + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=dmx), out=indptr[1:])
    return indptr, np.array([k[1] for k in keys], dtype=np.int32), np.array([cell[k] for k in keys], dtype=np.float32)


def device_mcl(indptr, indices, data, inflation, device=0, rounds=100):
    """the Markov loop on the GPU (libsohit so_mcl, csrc/mcl.hip) -> the final matrix as (indptr, indices, data) in the reference's
    storage order, stored zeros included.  No CPU path: raises when the HIP library or a device is missing."""
    import ctypes as C
    from . import _lib
    L = _lib.load()
    res = _lib.SoMclResult()
    ip = np.ascontiguousarray(indptr, dtype=np.int64)
    ix = np.ascontiguousarray(indices, dtype=np.int32)
    dv = np.ascontiguousarray(data, dtype=np.float32)
    rc = L.so_mcl(device, len(ip) - 1, ip.ctypes.data, ix.ctypes.data if len(ix) else None, dv.ctypes.data if len(dv) else None, float(inflation), int(rounds), 5,
                  1e-5, 1e-5, 1e-8, C.byref(res))
    if rc != 0:
        raise RuntimeError(L.so_mcl_last_error().decode())
    try:
        n, nnz = int(res.n), int(res.nnz)
        out_ip = np.ctypeslib.as_array(res.indptr, shape=(n + 1,)).copy()
        out_ix = np.ctypeslib.as_array(res.indices, shape=(max(nnz, 1),))[:nnz].copy()
        out_dv = np.ctypeslib.as_array(res.data, shape=(max(nnz, 1),))[:nnz].copy()
    finally:
        L.so_mcl_free(C.byref(res))
    return out_ip, out_ix, out_dv


def surviving_pairs(indptr, indices, data, prune=1e-5):
    """The reference's read-out of the final matrix (find_cluster.py:686-689): the coordinates of the entries that are not zero, paired
    IN ORDER with the raw data array -- which still holds the pruned zeros when the loop ran out of rounds, so the pairing can be
    shifted -- and kept where that value exceeds the threshold."""
    r, c = surviving_pair_columns(indptr, indices, data, prune)
    return list(zip(r.tolist(), c.tolist()))


def surviving_pair_columns(indptr, indices, data, prune=1e-5):
    """surviving_pairs as two arrays"""
    rows = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))
    nz = data != 0
    r, c = rows[nz], indices[nz]
    keep = data[:len(r)] > np.float32(prune)
    return r[keep].astype(np.int64), c[keep].astype(np.int64)


def mcl_block(edge_lines, inflation, mcl=device_mcl):
    """one batch of edges -> groups (lists of gene ids) in the reference's order"""
    names, indptr, indices, data = block_matrix(edge_lines)
    g = _Graph.from_pairs(surviving_pair_columns(*mcl(indptr, indices, data, inflation)))
    for comp in g.components():
        yield [names[e] for e in comp]


def mcl_block_arrays(gx, gy, z, gene_names, inflation, mcl=device_mcl):
    """mcl_block for a batch given as columns (see block_matrix_arrays)"""
    if len(gx) == 0:
        return
    names, indptr, indices, data = block_matrix_arrays(gx, gy, z, gene_names)
    g = _Graph.from_pairs(surviving_pair_columns(*mcl(indptr, indices, data, inflation)))
    for comp in g.components():
        yield [names[e] for e in comp]


def _component_labels(n, u, v):
    """connected components of an undirected graph on nodes 0 .. n-1: label = smallest node of the component (min-label propagation
    with pointer jumping)"""
    lab = np.arange(n, dtype=np.int64)
    if len(u) == 0:
        return lab
    while True:
        new = lab.copy()
        m = np.minimum(lab[u], lab[v])
        np.minimum.at(new, u, m)
        np.minimum.at(new, v, m)
        while True:
            nn = new[new]
            if np.array_equal(nn, new):
                break
            new = nn
        if np.array_equal(new, lab):
            return lab
        lab = new


def _numbered_components(pairs_u, pairs_v):
    """what `for flag, comp in enumerate(connected_components(G))` assigns after `G.add_edge(u, v)` for the pairs in order: nodes are
    known in order of first appearance (u before v), components are numbered in order of their first node.
    -> (node values in insertion order, component number of each of them)"""
    seq = np.empty(2 * len(pairs_u), dtype=np.int64)
    seq[0::2], seq[1::2] = pairs_u, pairs_v
    vals, first = np.unique(seq, return_index=True)
    order = np.argsort(first, kind='stable')
    nodes = vals[order]                                  # insertion order
    rank = np.empty(len(vals), dtype=np.int64)
    rank[order] = np.arange(len(vals))
    iu, iv = rank[np.searchsorted(vals, pairs_u)], rank[np.searchsorted(vals, pairs_v)]
    lab = _component_labels(len(nodes), iu, iv)
    roots = np.unique(lab)                               # ascending first node = discovery order
    return nodes, np.searchsorted(roots, lab)


def _edge_columns_native(data):
    """relation rows of a byte buffer through libsohit's tokeniser (so_tsv_*): -> (ids sorted bytewise, code of x, code of y, weight,
    weight text) of every row, or None when the library is missing, the input is tiny or a row is not what the fast path handles (the
    Python loop then decides, and raises what the reference would raise)"""
    import os
    n = data.count(b'\n')
    if n < int(os.environ.get('SOHIT_TSV_MIN', '4096')) or os.environ.get('SOHIT_TSV_NATIVE', '1') == '0':
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
    cols = np.array([0, 1, 2, 3, 2, 3], dtype=np.int32)
    numeric = np.array([0, 0, 0, 0, 1, 1], dtype=np.uint8)
    ntab = np.empty(n, dtype=np.int32)
    beg = np.empty((6, n), dtype=np.int64)
    ln = np.empty((6, n), dtype=np.int32)
    val = np.empty((6, n), dtype=np.float64)
    st = np.empty((6, n), dtype=np.uint8)
    if L.so_tsv_scan(ptr(buf), len(data), ptr(ls), n, 6, ptr(cols), ptr(numeric), ptr(ntab), ptr(beg), ptr(ln), ptr(val), ptr(st)) != 0:
        return None
    if np.any(ntab < 2):
        return None
    four = ntab == 3                                # `len(j) == 4`: a relation type in front of (x, y, weight)
    pick = lambda a, b: np.ascontiguousarray(np.where(four, a, b))
    xb, xl = pick(beg[1], beg[0]), pick(ln[1], ln[0])
    yb, yl = pick(beg[2], beg[1]), pick(ln[2], ln[1])
    zb, zl = pick(beg[3], beg[2]), pick(ln[3], ln[2])
    z, zs = pick(val[5], val[4]), pick(st[5], st[4])
    cx = np.empty(n, dtype=np.int64)
    cy = np.empty(n, dtype=np.int64)
    nb = np.empty(2 * n, dtype=np.int64)
    nlen = np.empty(2 * n, dtype=np.int32)
    nd = L.so_tsv_codes(ptr(buf), n, ptr(xb), ptr(xl), ptr(yb), ptr(yl), ptr(cx), ptr(cy), ptr(nb), ptr(nlen), 2 * n)
    if nd <= 0:
        return None
    use = cx <= cy                                  # `if x > y: continue` (ids compare like their bytes)
    if np.any(zs[use] != 0):
        return None                                 # a weight only Python's float() can judge
    names = [data[b:b + l].decode('utf-8') for b, l in zip(nb[:nd].tolist(), nlen[:nd].tolist())]
    u = np.flatnonzero(use)
    zb, zl = zb[u], zl[u]
    ztext = lambda i: data[int(zb[i]):int(zb[i]) + int(zl[i])]
    return names, cx[u], cy[u], z[u], ztext


def _edge_columns(lines):
    """relation rows ('[type\\t]x\\ty\\tweight\\n') -> (ids sorted bytewise, code of x, code of y, weight, weight text of row i) for the
    rows with x <= y, in file order"""
    data = None
    if hasattr(lines, 'read'):
        data = lines.buffer.read() if hasattr(lines, 'buffer') else lines.read()
    elif isinstance(lines, (bytes, bytearray)):
        data = bytes(lines)
    if data is not None:
        if isinstance(data, str):
            data = data.encode('utf-8')
        data = data.replace(b'\r\n', b'\n').replace(b'\r', b'\n')   # the reference reads in text mode (universal newlines)
        if data and not data.endswith(b'\n'):
            data = data[:-1] + b'\n'                                # ... and cuts the last character of every line, newline or not
        cols = _edge_columns_native(data)
        if cols is not None:
            return cols
        lines = data.decode('utf-8').splitlines(True)
    rows = [r for r in _rows(lines)]
    xs, ys, zs = [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows]
    both = np.array([g.encode('utf-8') for g in xs + ys], dtype=np.bytes_) if rows else np.zeros(0, dtype='S1')
    names, inv = np.unique(both, return_inverse=True)
    inv = inv.astype(np.int64)
    z = np.array([float(t) for t in zs], dtype=np.float64)
    return [g.decode('utf-8') for g in names.tolist()], inv[:len(rows)], inv[len(rows):], z, (lambda i: zs[i].encode('utf-8'))


def cnc(lines, inflation=1.5, chk=10 ** 7, mcl=device_mcl):
    """cnc (1470-1673): relation rows -> groups (lists of gene ids), in the reference's output order.  `lines`: an open file, bytes or an
    iterable of lines.  `mcl`: the Markov loop on a CSR block (the device implementation; the tests pass the scipy oracle to check this host
    bookkeeping on CPU).
    The reference keeps Python dictionaries and networkx graphs and moves the edges between its stages as sorted text files; here the
    rows are columns from the start (ids coded by their byte order, which is also the order `sort` compares them in), genes are integers
    numbered by first appearance (the insertion order of the reference's best-neighbour dictionary), and each of its orders --
    `popitem()` = last gene first, a graph's node order = first appearance in the `add_edge` sequence, components numbered by their
    first node, `LC_ALL=C sort -n` of the group-tagged edge lines -- is reproduced as index arithmetic."""
    names, cx, cy, Z, ztext = _edge_columns(lines)
    nrow = len(cx)
    if nrow == 0:
        return []
    # genes numbered by first appearance (x of a row before its y)
    seq = np.empty(2 * nrow, dtype=np.int64)
    seq[0::2], seq[1::2] = cx, cy
    vals, first = np.unique(seq, return_index=True)
    order = np.argsort(first, kind='stable')
    number_of_code = np.zeros(len(names), dtype=np.int64)
    number_of_code[vals[order]] = np.arange(len(vals))
    X, Y = number_of_code[cx], number_of_code[cy]
    gene_names = [names[c] for c in vals[order].tolist()]
    n = len(vals)
    # level 1: every gene linked to its best-scoring neighbour(s); the dictionary is emptied last gene first, a gene's ties in file order
    best = np.full(n, -np.inf)
    np.maximum.at(best, X, Z)
    np.maximum.at(best, Y, Z)
    a = np.concatenate([X, Y])
    b = np.concatenate([Y, X])
    ridx = np.concatenate([np.arange(nrow) * 2, np.arange(nrow) * 2 + 1])   # (x, y) of a row before its (y, x)
    tie = np.concatenate([Z, Z]) == best[a]
    a, b, ridx = a[tie], b[tie], ridx[tie]
    o = np.lexsort((ridx, -a))
    genes1, comp1_of_node = _numbered_components(a[o], b[o])
    comp1 = np.zeros(n, dtype=np.int64)
    comp1[genes1] = comp1_of_node
    # level 2: components joined by an edge -- seen only when BOTH component numbers are non-zero
    cX, cY = comp1[X], comp1[Y]
    m = (cX != 0) & (cY != 0)
    k0, k1 = np.minimum(cX[m], cY[m]), np.maximum(cX[m], cY[m])
    nc = int(comp1.max()) + 2
    key = k0 * nc + k1
    _, first = np.unique(key, return_index=True)
    first.sort()
    nodes2, comp2_of_node = _numbered_components(k0[first], k1[first])
    group_of_comp = np.full(nc, -1, dtype=np.int64)
    group_of_comp[nodes2] = comp2_of_node
    grp = group_of_comp[comp1]
    # edges inside one level-2 group whose number is non-zero (-1, the pool of unmerged components, included)
    gx, gy = grp[X], grp[Y]
    keep = np.flatnonzero((gx != 0) & (gy != 0) & (gx == gy))
    if len(keep) == 0:
        return []
    # LC_ALL=C sort -n of the lines 'group\tx\ty\tweight': the leading number, then the whole line bytewise = (x, y, weight text)
    # bytewise (ids hold no byte below the tab, so a shorter id sorts before its extensions, like its code does)
    kg, kx, ky = gx[keep], cx[keep], cy[keep]
    order = np.lexsort((ky, kx, kg))
    sg, sx, sy = kg[order], kx[order], ky[order]
    dup = np.flatnonzero((sg[1:] == sg[:-1]) & (sx[1:] == sx[:-1]) & (sy[1:] == sy[:-1]))
    if len(dup):                                          # the same pair twice in a group: its lines are ordered by the weight's text
        run_start = dup[np.concatenate([[True], np.diff(dup) > 1])]
        for s0 in run_start.tolist():
            e0 = s0 + 1
            while e0 < len(order) and sg[e0] == sg[s0] and sx[e0] == sx[s0] and sy[e0] == sy[s0]:
                e0 += 1
            seg = order[s0:e0].tolist()
            seg.sort(key=lambda r: ztext(int(keep[r])) + b'\n')
            order[s0:e0] = seg
    rows_sorted = keep[order]
    kcls = kg[order]
    # batches: a new one starts where the group changes once the running batch holds more than chk rows
    change = np.concatenate([[0], np.flatnonzero(kcls[1:] != kcls[:-1]) + 1, [len(kcls)]])
    out, start = [], 0
    for c in change[1:-1].tolist():
        if c - start > chk:
            out.extend(mcl_block_arrays(X[rows_sorted[start:c]], Y[rows_sorted[start:c]], Z[rows_sorted[start:c]], gene_names, inflation, mcl))
            start = c
    out.extend(mcl_block_arrays(X[rows_sorted[start:]], Y[rows_sorted[start:]], Z[rows_sorted[start:]], gene_names, inflation, mcl))
    return out


def parse(argv):
    from .fsearch import parse_flags
    return parse_flags(argv, DEFAULTS)


def main(argv=None):
    argv = list(sys.argv if argv is None else argv)
    args = parse(argv)
    if args['-i'] == '':
        manual_print(argv[0] if argv else 'find_cluster.py')
        raise SystemExit()
    try:
        qry, ifl, alg = args['-i'], float(args['-I']), args['-a'].lower()
        float(args['-d']), float(args['-p']), int(args['-t']), int(args['-b'])
    except Exception:
        manual_print(argv[0] if argv else 'find_cluster.py')
        raise SystemExit()
    if alg != 'mcl':
        sys.stderr.write('find_cluster: only -a mcl is provided (the reference\'s affinity-propagation modes are not)\n')
        return 2
    # the HIP runtime and the Markov kernels' code object take ~0.25 s to come up: a thread brings them up on a 1 x 1 matrix while
    # this one reads the edges and finds the components (failures there are left to the real call, which reports them)
    import threading

    def warm():
        try:
            device_mcl(np.array([0, 1]), np.array([0]), np.array([1.], dtype=np.float32), ifl, rounds=1)
        except Exception:
            pass
    wt = threading.Thread(target=warm, daemon=True)
    wt.start()
    try:
        with open(qry, 'r') as f:
            groups = cnc(f, ifl)
        w = sys.stdout.write
        for grp in groups:
            w('\t'.join(grp) + '\n')
    finally:
        wt.join(30.0)   # never leave the interpreter while the thread is inside HIP initialisation (an edge-less input ends before it)
    return 0


if __name__ == '__main__':
    if __package__ in (None, ''):
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from swiftortho_amd.find_cluster import main as _m
        sys.exit(_m())
    sys.exit(main())

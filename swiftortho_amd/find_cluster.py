"""find_cluster -- counterpart of SwiftOrtho's bin/find_cluster.py for `-a mcl`: from the orthology
relations file (find_orth output) to ortholog groups, one tab-separated group per line, same flags,
same stdout.

Last stage of BASELINE config 5 (find_hit -> find_orth -> find_cluster -a mcl -I 1.5); a SURVEY.md 8f
"next" row.  Host code: the graphs here have a few edges per protein, the stage is bounded by text
parsing, and the Markov-cluster iteration runs on float32 CSR matrices through scipy's SpGEMM exactly
as the reference's does (bit-identical expansion is what keeps the pruning decisions, and therefore
the groups, identical).  Only `mcl` is provided; the reference's affinity-propagation modes
(`-a apc|sap`) are refused with a message.

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


def _normalize_columns(x):
    """normalize (636-646): column sums in float32; the epsilon rule as written"""
    y = np.asarray(x.sum(0))[0]
    if y.min() == 0 and y.max() > 0:
        y += y.nonzero()[0].min() / 1e3
    else:
        y += 1e-8
    x.data /= y.take(x.indices, mode='clip')


def markov_cluster(x, inflation=1.5, expansion=2, prune=1e-5, rtol=1e-5, atol=1e-8, rounds=100, check=5):
    """mcl (652-689) on a scipy csr_matrix (float32) -> list of (row, col) pairs that survive"""
    for i in range(rounds):
        _normalize_columns(x)
        if i % check == 0:
            x_old = x.copy()
        x = x ** expansion          # spmatrix power = matrix product (SpGEMM; entries that sum to zero are not stored)
        x.data **= inflation
        if i % check == 0 and i > 0:
            if (abs(x - x_old) - rtol * abs(x_old)).max() <= atol:
                break
        x.data[x.data < prune] = 0.
    rows, cols = x.nonzero()
    vals = x.data                   # NOT masked: misaligned with (rows, cols) when pruned zeros are still stored
    return [(int(r), int(c)) for r, c, k in zip(rows, cols, vals) if k > prune]


def mcl_block(edge_lines, inflation):
    """mcl_xyz (1425-1467): lines 'x\\ty\\tz...' of one batch -> groups (lists of ids) in the reference's order"""
    from scipy import sparse
    l2n = {}
    for i in edge_lines:
        x, y = i.split('\t', 3)[:2]
        if x not in l2n:
            l2n[x] = len(l2n)
        if y not in l2n:
            l2n[y] = len(l2n)
    dmx = len(l2n) + 1
    cell = {}
    for i in edge_lines:
        x, y, z = i.split('\t', 4)[:3]
        if x > y:
            continue
        X, Y = l2n[x], l2n[y]
        Z = np.float32(float(z))
        cell[(X, Y)] = Z
        cell[(Y, X)] = Z
        if cell.get((X, X), np.float32(0)) < Z:
            cell[(X, X)] = Z
        if cell.get((Y, Y), np.float32(0)) < Z:
            cell[(Y, Y)] = Z
    n2l = {}
    while l2n:
        key, val = l2n.popitem()
        n2l[val] = key
    keys = sorted(cell)             # row-major, columns ascending: the canonical CSR of the reference's lil -> csr conversion
    r = np.fromiter((k[0] for k in keys), dtype=np.int32, count=len(keys))
    c = np.fromiter((k[1] for k in keys), dtype=np.int32, count=len(keys))
    v = np.fromiter((cell[k] for k in keys), dtype=np.float32, count=len(keys))
    keep = v != 0                   # lil_matrix does not store assigned zeros
    m = sparse.csr_matrix((v[keep], (r[keep], c[keep])), shape=(dmx, dmx), dtype='float32')
    g = _Graph()
    for a, b in markov_cluster(m, inflation):
        g.add_edge(a, b)
    for comp in g.components():
        yield [n2l[e] for e in comp]


def cnc(lines, inflation=1.5, chk=10 ** 7):
    """cnc (1470-1673): relation rows -> groups (lists of gene ids), in the reference's output order"""
    lines = list(lines)
    nns = {}
    for x, y, z in _rows(lines):
        Z = float(z)
        for a, b in ((x, y), (y, x)):
            if a in nns:
                if Z > nns[a][0]:
                    nns[a] = [Z, b]
                elif Z == nns[a][0]:
                    nns[a].append(b)
            else:
                nns[a] = [Z, b]
    g = _Graph()
    while nns:
        x, j = nns.popitem()
        for y in j[1:]:
            g.add_edge(x, y)
    l2n = {}
    for flag, comp in enumerate(g.components()):
        for j in comp:
            l2n[j] = flag
    g2 = _Graph()
    seen_keys = set()
    for x, y, z in _rows(lines):
        X, Y = l2n.get(x), l2n.get(y)
        if X and Y:
            key = (X, Y) if X < Y else (Y, X)
            if key not in seen_keys:
                seen_keys.add(key)
                g2.add_edge(key[0], key[1])
    n2n = {}
    for flag, comp in enumerate(g2.components()):
        for j in comp:
            n2n[j] = flag
    for i in l2n:
        l2n[i] = n2n.get(l2n[i], -1)
    kept = []
    for x, y, z in _rows(lines):
        cx, cy = l2n.get(x), l2n.get(y)
        if cx and cy and cx == cy:
            kept.append((cx, ('\t'.join(map(str, [cx, x, y, z])) + '\n')))
    kept.sort(key=lambda t: (t[0], t[1].encode('latin-1')))   # LC_ALL=C sort -n: leading number, then the whole line bytewise
    out, batch, cls, flag = [], [], None, 0
    for cx, line in kept:
        c = line.split('\t', 2)[0]
        if c != cls:
            if flag > chk:
                out.extend(mcl_block(batch, inflation))
                batch, flag = [], 0
            cls = c
        batch.append(line.split('\t', 1)[1])
        flag += 1
    out.extend(mcl_block(batch, inflation))
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
    with open(qry, 'r') as f:
        groups = cnc(f, ifl)
    w = sys.stdout.write
    for grp in groups:
        w('\t'.join(grp) + '\n')
    return 0


if __name__ == '__main__':
    if __package__ in (None, ''):
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from swiftortho_amd.find_cluster import main as _m
        sys.exit(_m())
    sys.exit(main())

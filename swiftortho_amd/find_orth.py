"""find_orth -- counterpart of SwiftOrtho's bin/find_orth.py: from a find_hit .sc file to the
orthology relations file (`IP` in-paralogs, `OT` orthologs, `CO` co-orthologs; OrthoMCL-style
normalised scores), same flags, same stdout, line for line.

This stage is text and dictionary work on a few bytes per reported hit; it stays on the host
(the reference does it with GNU sort + mmap'd binary searches over temp files; here everything is
in memory and nothing is written next to the input).  It is a SURVEY.md 8f "next" row: the
consumer of the file the GPU search writes.

Reference behaviour reproduced (bin/find_orth.py), including what is arguably accidental:
  * rows are grouped by CONSECUTIVE query id; per (query, subject) the best normalised score wins
    (strictly greater replaces); filters: query coverage (1 + |qed - qst|) / qlen >= -c, identity >= -y;
    rows whose numeric columns do not parse are skipped (blastparse, 158-234);
  * -n no | bsr (score / first score seen for the query) | bal (score / alignment length);
  * per query: best score per subject taxon, best out-of-taxon score; same-taxon hits >= that and not
    the query itself -> in-paralog candidates (both orientations), other-taxon hits equal to the taxon's
    best -> ortholog candidates, the rest -> co-ortholog candidates (get_qIPO, 298-348);
  * a candidate pair becomes a relation only when it was proposed from both sides: exactly TWO lines
    in the byte-sorted candidate file (three or more: dropped); score = mean of the two -- except for
    the LAST pair of the file, which gets the max (get_IPO, 351-377);
  * in-paralog normalisation: per taxon, mean over pairs with an ortholog on either side, else over all
    pairs (507-543); printed for qid < sid only (627-649);
  * co-orthologs: for every ortholog pair, every (in-paralog-or-self of q) x (in-paralog-or-self of s)
    combination that is a co-ortholog CANDIDATE in that orientation, with the best candidate score; the
    pair itself is looked up too (the reference compares bytes with str there, 586); needs at least one
    in-paralog on either side (547-617);
  * OT / CO normalisation: consecutive lines of one query taxon form a block, duplicates inside a block
    dropped -- except that the block's first pair is not registered (`set((qid, sid))`), so one repeat
    of it survives; score / mean score of the block's pairs with the same subject taxon (681-762).
Scores print as Python's str(float), like the reference under Python 3.
"""
import sys

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


def blastparse(lines, coverage=.5, identity=0., norm='no', sep='|'):
    """find_orth.py:158-234 -> lists of [qid, sid, Score], one list per run of rows with the same query id"""
    output = {}
    len_dict = {}
    flag = None
    mbsc_dict = {}
    for i in lines:
        j = i[:-1].split('\t')
        qid, sid = j[:2]
        assert sep in qid and sep in sid
        try:
            idy, aln, mis, gop, qst, qed, sst, sed, evalue, score = list(map(float, j[2:12]))
        except Exception:
            continue
        if len(j) > 13:
            try:
                qln, sln = list(map(float, j[12:14]))
            except Exception:
                continue
        else:
            if qid in len_dict:
                qln = len_dict[qid]
            else:
                qln = max(qst, qed)
                len_dict[qid] = qln
        qcv = (1. + abs(qed - qst)) / qln
        if qcv < coverage or idy < identity:
            continue
        if norm == 'bsr':
            if qid not in mbsc_dict:
                mbsc_dict[qid] = score
            Score = score / mbsc_dict[qid]
        elif norm == 'bal':
            Score = score / aln
        else:
            Score = score
        if flag != qid:
            if output:
                yield list(output.values())
            output = {}
            flag = qid
            output[sid] = [qid, sid, Score]
        elif sid not in output or output[sid][-1] < Score:
            output[sid] = [qid, sid, Score]
    if output:
        yield list(output.values())


def get_qIPO(hits, sep='|'):
    """find_orth.py:298-348 -> candidate lines (ips in both orientations, ots, cos) of one query"""
    sco_max = {}
    out_max = 0
    for qid, sid, sco in hits:
        qtx, stx = qid.split(sep)[0], sid.split(sep)[0]
        sco_max[stx] = max(sco_max.get(stx, 0), sco)
        if qtx != stx:
            out_max = max(out_max, sco)
    visit = set()
    ips, ots, cos = [], [], []
    for qid, sid, sco in hits:
        if sid in visit:
            continue
        visit.add(sid)
        qtx, stx = qid.split(sep)[0], sid.split(sep)[0]
        if not qid < sid:
            qid, sid = sid, qid
        if qtx == stx:
            if sco >= out_max and qid != sid:
                ips.append((qid, sid, sco))
                ips.append((sid, qid, sco))
        elif sco >= sco_max[stx]:
            ots.append((qid, sid, sco))
        else:
            cos.append((qid, sid, sco))
    return ips, ots, cos


def _sorted_lines(cands):
    """`LC_ALL=C sort` of the candidate file: whole lines 'a\\tb\\tstr(score)\\n' compared bytewise"""
    keyed = [(("%s\t%s\t%s\n" % (a, b, str(s))).encode('latin-1'), a, b, s) for a, b, s in cands]
    keyed.sort(key=lambda t: t[0])
    return keyed


def get_IPO(sorted_cands):
    """find_orth.py:351-377 over the sorted candidate lines -> (qid, sid, score) of the pairs proposed exactly twice"""
    out = []
    n = len(sorted_cands)
    k = 0
    while k < n:
        a, b = sorted_cands[k][1], sorted_cands[k][2]
        e = k
        while e < n and sorted_cands[e][1] == a and sorted_cands[e][2] == b:
            e += 1
        if e - k == 2:
            s0, s1 = sorted_cands[k][3], sorted_cands[k + 1][3]
            out.append((a, b, max(s0, s1) if e == n else sum([s0, s1]) / 2.))
        k = e
    return out


def _same_taxon_blocks(pairs, sep):
    """get_sam_tax (find_orth.py:681-702)"""
    flag, out, visit = None, [], set()
    for qid, sid, sco in pairs:
        qtx = qid.split(sep)[0]
        if qtx != flag:
            if out:
                yield out
            flag = qtx
            out = [[qid, sid, sco]]
            visit = set((qid, sid))   # as in the reference: the two ids, not the pair
        elif (qid, sid) not in visit:
            out.append([qid, sid, sco])
            visit.add((qid, sid))
    if out:
        yield out


def _normalised(block, sep):
    """n_co_ot (find_orth.py:728-746)"""
    avgs = {}
    for qid, sid, sco in block:
        stx = sid.split(sep)[0]
        if stx in avgs:
            avgs[stx][0] += sco
            avgs[stx][1] += 1.
        else:
            avgs[stx] = [sco, 1.]
    for k in avgs:
        a, b = avgs[k]
        avgs[k] = a / b
    for qid, sid, sco in block:
        yield qid, sid, sco / avgs[sid.split(sep)[0]]


def find_orth(lines, coverage=.5, identity=0., norm='no', sep='|'):
    """.sc rows (iterable of text lines) -> list of output lines ('IP|OT|CO\\tqid\\tsid\\tscore'), in the reference's order"""
    qips, qots, qcos = [], [], []
    for hits in blastparse(lines, coverage, identity, norm, sep):
        a, b, c = get_qIPO(hits, sep)
        qips += a
        qots += b
        qcos += c
    # orthologs
    ots = get_IPO(_sorted_lines(qots))
    inots = set()
    for qid, sid, sco in ots:
        inots.add(qid)
        inots.add(sid)
    # in-paralogs and their per-taxon normalisers
    ips = get_IPO(_sorted_lines(qips))
    ipqa, IPqA = {}, {}
    for qid, sid, sco in ips:
        qtx = qid.split(sep)[0]
        if qid < sid:
            if qid in inots or sid in inots:
                if qtx in ipqa:
                    ipqa[qtx][0] += float(sco)
                    ipqa[qtx][1] += 1.
                else:
                    ipqa[qtx] = [float(sco), 1.]
            if qtx in IPqA:
                IPqA[qtx][0] += float(sco)
                IPqA[qtx][1] += 1.
            else:
                IPqA[qtx] = [float(sco), 1.]
    for k in IPqA:
        a, b = ipqa[k] if k in ipqa else IPqA[k]
        IPqA[k] = a / b
    # co-orthologs
    cos = []
    if ips and qcos:
        ip_of = {}
        for qid, sid, sco in ips:     # IPs.txt is sorted: partners come out in that order
            ip_of.setdefault(qid, []).append(sid)
        co_best = {}
        co_first = {}
        for _, a, b, s in _sorted_lines(qcos):
            if (a, b) not in co_best:
                co_best[(a, b)] = s
                co_first[(a, b)] = (a, b)
            elif s > co_best[(a, b)]:
                co_best[(a, b)] = s
        for qid, sid, sco in ots:
            qp, sp = ip_of.get(qid, []), ip_of.get(sid, [])
            if not qp and not sp:
                continue
            visit = set()
            for qip in qp + [qid]:
                for sip in sp + [sid]:
                    if (qip, sip) in visit:
                        continue
                    visit.add((qip, sip))
                    if (qip, sip) in co_best:
                        cos.append((qip, sip, co_best[(qip, sip)]))
    out = []
    for qid, sid, score in ips:
        if qid >= sid:
            continue
        avg = IPqA[qid.split(sep)[0]]
        try:
            out.append('\t'.join(map(str, ['IP', qid, sid, float(score) / avg])))
        except Exception:
            continue
    for kind, pairs in (('OT', ots), ('CO', cos)):
        for block in _same_taxon_blocks(pairs, sep):
            for j in _normalised(block, sep):
                out.append(kind + '\t' + '\t'.join(map(str, j)))
    return out


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
    with open(qry, 'r') as f:
        lines = find_orth(f, coverage, identity, norm, sep)
    w = sys.stdout.write
    for l in lines:
        w(l + '\n')
    return 0


if __name__ == '__main__':
    if __package__ in (None, ''):
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from swiftortho_amd.find_orth import main as _m
        sys.exit(_m())
    sys.exit(main())

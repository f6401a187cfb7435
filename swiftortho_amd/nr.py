"""Exact-duplicate collapse / expand around the search: counterparts of SwiftOrtho's
scripts/nr_flt.py and scripts/nr2full.py, the two helpers scripts/run_all_fast.py wraps around find_hit
(109-118): identical sequences are searched once and the hits are multiplied back afterwards.

nr_flt (scripts/nr_flt.py:1-27): records with the same residue string are merged in order of first
appearance; the merged header is the record ids (header up to the first whitespace) joined by ';;;';
the sequence is printed on one line.  The reference parses with Bio.SeqIO (absent from this image), so
this half is restated from the source and checked by hand-written cases only.

nr2full (scripts/nr2full.py:21-44): every row of the collapsed search is expanded to the cross product
of the ';;;'-separated query ids and subject ids; columns 3..14 are kept, the last two columns (query
ordinal, subject header) are replaced by the individual query and subject ids; inside a run of rows
with the same collapsed query the expanded rows come out grouped by individual query, groups in order
of first appearance.  Pinned by running the reference script (stdlib only) in the build container.

Host text utilities: nothing here touches the device.
"""
import sys


def fasta_records(lines):
    """(header without '>', sequence) per record; sequence lines stripped and joined, as Bio.SeqIO's FASTA parser does"""
    head, seq = None, []
    for line in lines:
        if line.startswith('>'):
            if head is not None:
                yield head, ''.join(seq)
            head, seq = line[1:].rstrip('\r\n'), []
        elif head is not None:
            seq.append(''.join(line.split()))
    if head is not None:
        yield head, ''.join(seq)


def nr_flt(lines):
    """FASTA lines -> output lines of nr_flt.py"""
    groups = {}
    for head, seq in fasta_records(lines):
        parts = head.split(None, 1)
        rid = parts[0] if parts else ''       # SeqRecord.id: the first word of the title
        groups.setdefault(seq, []).append(rid)
    out = []
    for seq, ids in groups.items():
        out.append('>' + ';;;'.join(ids))
        out.append(seq)
    return out


def nr2full(lines):
    """.sc rows of the collapsed search -> expanded rows"""
    out = []

    def flush(hits):
        outs = {}
        for j in hits:
            qds, rds = j[:2]
            for qd in qds.split(';;;'):
                for rd in rds.split(';;;'):
                    q = qd.split(' ')[0]
                    r = rd.split(' ')[0]
                    outs.setdefault(q, []).append('\t'.join([q, r] + j[2:-2] + [qd, rd]))
        for vals in outs.values():
            out.extend(vals)

    hits = []
    for i in lines:
        j = i[:-1].split('\t')
        if hits and hits[0][0] != j[0]:
            flush(hits)
            hits = [j]
        else:
            hits.append(j)
    if hits:
        flush(hits)
    return out


def main_nr_flt(argv=None):
    argv = list(sys.argv if argv is None else argv)
    f = open(argv[1], 'r') if len(argv) > 1 else sys.stdin
    for l in nr_flt(f):
        print(l)
    return 0


def main_nr2full(argv=None):
    argv = list(sys.argv if argv is None else argv)
    if len(argv) < 2:
        print('python this.py foo.sc')
        return 0
    with open(argv[1], 'r') as f:
        for l in nr2full(f):
            print(l)
    return 0

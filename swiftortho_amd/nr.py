"""Exact-duplicate collapse / expand around the search: counterparts of SwiftOrtho's
scripts/nr_flt.py and scripts/nr2full.py, the two helpers scripts/run_all_fast.py wraps around find_hit
(run_all_fast.py:109-118): identical sequences are searched once and the hits are multiplied back afterwards.
`search_collapsed()` is that wrapper: nr_flt -> find_hit -> nr2full with the reference's file names.

nr_flt (scripts/nr_flt.py:1-27): records with the same residue string are merged in order of first
appearance; the merged header is the record ids joined by ';;;'; the sequence is printed on one line.
The reference reads the file with Bio.SeqIO.parse(.., 'fasta') (Biopython is absent from this image): the
record rules below are Biopython's documented ones -- title = the '>' line without '>' and trailing blanks,
id = its first word, sequence = the following lines right-stripped and joined with blanks and carriage
returns removed, text before the first '>' ignored.  Pinned by the golden `nr_dups.nr.fsa` = stdout of the
REAL nr_flt.py on plain FASTA (where every parser agrees; tools/refharness/make_nr_goldens.py); the handling
of inner blanks / CR / leading text follows the documented rules and is NOT pinned by a run of Biopython.

nr2full (scripts/nr2full.py:14-44): every row of the collapsed search is expanded to the cross product of the
';;;'-separated query ids and subject ids; columns 3..14 are kept, the last two columns (query ordinal,
subject header) are replaced by the individual query and subject names; inside a run of rows with the same
collapsed query the expanded rows come out grouped by individual query, groups in order of first
appearance.  Pinned by the golden `nr_dups.full.sc` = stdout of the REAL nr2full.py.

Host text utilities: nothing here touches the device.
"""
import itertools
import os
import sys

JOIN = ';;;'


def fasta_records(lines):
    """(title, sequence) per record, by Biopython's FASTA rules (see the module docstring)"""
    title, seq = None, []
    for line in lines:
        if line[:1] == '>':
            if title is not None:
                yield title, ''.join(seq).replace(' ', '').replace('\r', '')
            title, seq = line[1:].rstrip(), []
        elif title is not None:
            seq.append(line.rstrip())
    if title is not None:
        yield title, ''.join(seq).replace(' ', '').replace('\r', '')


def nr_flt(lines):
    """FASTA lines -> output lines of nr_flt.py: one '>id;;;id...' + residue line per distinct sequence"""
    members = {}
    for title, seq in fasta_records(lines):
        words = title.split(None, 1)
        members.setdefault(seq, []).append(words[0] if words else '')
    return [l for seq, ids in members.items() for l in ('>' + JOIN.join(ids), seq)]


def nr2full(lines):
    """.sc rows of the collapsed search -> expanded rows (no newline)"""
    out = []
    rows = (l[:-1].split('\t') for l in lines)
    for _, run in itertools.groupby(rows, key=lambda c: c[0]):
        expanded, rank = [], {}
        for c in run:
            mid = c[2:-2]
            for qname in c[0].split(JOIN):
                q = qname.split(' ')[0]
                k = rank.setdefault(q, len(rank))
                for sname in c[1].split(JOIN):
                    expanded.append((k, '\t'.join([q, sname.split(' ')[0]] + mid + [qname, sname])))
        expanded.sort(key=lambda t: t[0])   # stable: rows of one individual query stay in generation order
        out.extend(t[1] for t in expanded)
    return out


def search_collapsed(fas, seed='1111111', cpus='1', hits='1000', device_flags=(), python=sys.executable):
    """run_all_fast.py:95-118 -- `<fas>_nr.fsa` (collapsed proteome), `<fas>_results/<name>_nr.fsa.sc` (its self-search:
    bin/find_hit.py -e 1e-5 -m 5e-2 -s seed -a cpus -v hits), `<fas>_results/<name>.sc` (hits multiplied back).
    Returns the path of the last file."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = fas.split(os.sep)[-1]
    res = fas + '_results'
    os.makedirs(res, exist_ok=True)
    nr_fa = fas + '_nr.fsa'
    with open(fas, 'r') as f, open(nr_fa, 'w') as o:
        for l in nr_flt(f):
            o.write(l + '\n')
    nr_sc = os.path.join(res, name + '_nr.fsa.sc')
    cmd = [python, os.path.join(root, 'bin', 'find_hit.py'), '-p', 'blastp', '-i', nr_fa, '-d', nr_fa, '-o', nr_sc, '-e', '1e-5', '-s', seed,
           '-m', '5e-2', '-a', str(cpus), '-v', str(hits)] + list(device_flags)
    with open(os.path.join(res, 'log'), 'w') as log:
        rc = subprocess.run(cmd, stdout=log, stderr=subprocess.STDOUT).returncode
    if rc != 0 or not os.path.isfile(nr_sc):
        raise RuntimeError('find_hit failed (exit %d): see %s' % (rc, os.path.join(res, 'log')))
    full = os.path.join(res, name + '.sc')
    with open(nr_sc, 'r') as f, open(full, 'w') as o:
        for l in nr2full(f):
            o.write(l + '\n')
    return full


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

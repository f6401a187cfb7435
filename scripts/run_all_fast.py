#!/usr/bin/env python3
"""Counterpart of the search + orthology + clustering steps of SwiftOrtho's scripts/run_all_fast.py (95-193): duplicate
collapse, all-vs-all search on the GPU, expansion, find_orth, clustering.  Same flags for those steps (-i -s -a -v -c -y -n -A -I),
same files under <fasta>_results/ (.sc, .opc, .xyz, .clsr); the pan-genome / species-tree / operon steps of the reference (external
tools: trimal, fasttree) are outside SURVEY.md section 8 and are not run.

Clustering step, as the reference has it (139-193): the gene ids of the .opc file are recoded to decimal numbers by first appearance
(<name>.xyz, three columns), THAT file is clustered, and the numbers are mapped back into <name>.clsr (the .grp file in between is
removed).  The recode matters: find_cluster compares ids as strings (`if x > y: continue`, `sort`), and '5' > '12'.
What is NOT the reference here: for `-A mcl` -- its default -- the reference wrapper runs the external `mcl` PROGRAM on the .xyz file
(`mcl ... --abc -te N -I 1.5`; van Dongen's binary, not part of the reference repository and absent from this image).  This wrapper runs
`bin/find_cluster.py -a mcl` (the reference's own Python implementation of the algorithm, SURVEY.md 8f-2, Markov loop on the GPU) on the
same .xyz instead: the two implement the same algorithm with different pruning schemes, so the .clsr of `-A mcl` is not pinned against
the reference wrapper's.  `-A apc` / `-A sap` are refused by find_cluster here (affinity propagation is outside 8f-2)."""
import os
import subprocess
import sys
from time import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swiftortho_amd import nr  # noqa: E402
from swiftortho_amd.fsearch import parse_flags  # noqa: E402

ARGS = {'-i': '', '-r': '', '-p': '', '-s': '1111111', '-c': '.5', '-y': '50', '-n': 'no', '-l': '.05', '-u': '.95', '-a': '1', '-A': 'mcl',
        '-I': '1.5', '-v': '1000'}


def main(argv):
    a = parse_flags(argv, ARGS)
    if a['-i'] == '':
        print('  python %s -i foo.pep.fsa [-s seed] [-a gpus] [-v hits] [-c cov] [-y identity] [-n no|bsr|bal] [-A mcl] [-I 1.5]' % argv[0])
        raise SystemExit()
    fas, name = a['-i'], a['-i'].split(os.sep)[-1]
    t = time()
    sc = nr.search_collapsed(fas, a['-s'], a['-a'], a['-v'])
    print('all to all homologous searching time:', time() - t)
    t = time()
    opc = '%s_results/%s.opc' % (fas, name)
    with open(opc, 'wb') as o:
        subprocess.run([sys.executable, os.path.join(ROOT, 'bin', 'find_orth.py'), '-i', sc, '-c', a['-c'], '-y', a['-y'], '-n', a['-n'], '-t', 'y'],
                       stdout=o, check=True)
    print('orthomcl algorithm time:', time() - t)
    t = time()
    # gene ids -> numbers by first appearance (run_all_fast.py:143-160)
    xyz, grp, clsr = ('%s_results/%s.%s' % (fas, name, e) for e in ('xyz', 'grp', 'clsr'))
    id2n = {}
    with open(opc) as f, open(xyz, 'w') as o:
        for line in f:
            typ, qid, sid, sco = line.split('\t')
            for g in (qid, sid):
                if g not in id2n:
                    id2n[g] = len(id2n)
            o.write('%d\t%d\t%s' % (id2n[qid], id2n[sid], sco))
    with open(grp, 'wb') as o:
        subprocess.run([sys.executable, os.path.join(ROOT, 'bin', 'find_cluster.py'), '-i', xyz, '-a', a['-A'], '-I', a['-I']], stdout=o, check=True)
    # numbers -> gene ids (180-190)
    n2id = {str(n): g for g, n in id2n.items()}
    with open(grp) as f, open(clsr, 'w') as o:
        for line in f:
            o.write('\t'.join(n2id[k] for k in line[:-1].split('\t')) + '\n')
    os.remove(grp)
    print('use %s to group protein family time:' % a['-A'], time() - t)
    return 0


if __name__ == '__main__':
    sys.exit(main(list(sys.argv)))

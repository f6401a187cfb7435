#!/usr/bin/env python3
"""Drop-in for the search + orthology + clustering steps of SwiftOrtho's scripts/run_all_fast.py (95-193): duplicate
collapse, all-vs-all search on the GPU, expansion, find_orth, find_cluster -a mcl.  Same flags for those steps
(-i -s -a -v -c -y -n -A -I), same file names under <fasta>_results/; the pan-genome / species-tree / operon steps of the
reference (external tools: mcl, trimal, fasttree) are outside SURVEY.md section 8 and are not run."""
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
    clsr = '%s_results/%s.clsr' % (fas, name)
    with open(clsr, 'wb') as o:
        subprocess.run([sys.executable, os.path.join(ROOT, 'bin', 'find_cluster.py'), '-i', opc, '-a', a['-A'], '-I', a['-I']], stdout=o, check=True)
    print('use %s to group protein family time:' % a['-A'], time() - t)
    return 0


if __name__ == '__main__':
    sys.exit(main(list(sys.argv)))

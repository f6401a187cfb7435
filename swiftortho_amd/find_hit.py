"""find_hit -- drop-in for SwiftOrtho's bin/find_hit.py (-p blastp): same flag grammar, defaults,
output file and row format, with the search done by libsohit.so on MI355X GPUs.

Reference behaviour mirrored (bin/find_hit.py:194-358):
  * `-k v` or `-kv`; unknown tokens skipped; defaults of find_hit.py:227-228
    (-v 500 -s 11111111 -e 1e-3 -m 1e-3 -t -1 -r aa9 -j 1 -F T -O wb -M 120000000 -c 50000 -a 1);
  * `-p` must be blastp and -i/-d non-empty, else the manual is printed and the program exits;
  * `-r aa9|aa20|custom`; chunk size = int(-c / number of '/'-separated alphabets) (273-274);
  * output rows in ascending query order (the reference concatenates its per-block part files in
    block order, 135-146).
Differences by design: `-a` is the number of GPUs (one process per GPU, queries sharded, hit
records gathered over RCCL) instead of CPU worker processes; nothing is spilled to `-T`.
Not implemented yet: the >= 4.2e9-byte reference split/merge (303-351) -- such inputs are refused.
"""
import os
import subprocess
import sys

AA9 = 'AST,CFILMVY,DN,EQ,G,H,KR,P,W'
AA20 = 'A,S,T,C,F,I,L,M,V,Y,D,N,E,Q,G,H,K,R,P,W'

DEFAULTS = {'-p': '', '-v': '500', '-s': '11111111', '-i': '', '-d': '', '-e': '1e-3', '-l': '-1', '-u': '-1', '-m': '1e-3',
            '-t': '-1', '-r': 'aa9', '-j': '1', '-F': 'T', '-o': '', '-D': '', '-O': 'wb', '-L': '-1', '-U': '-1',
            '-M': '120000000', '-c': '50000', '-a': '1', '-T': ''}


def manual_print(prog='find_hit.py'):
    print('Usage:')
    print('  search:')
    print('    python %s -p blastp -i qry.fsa -d db.fsa' % prog)
    print('Parameters:')
    for line in ('-p: program', '-i: query sequences in fasta format', '-l: start index of query sequences',
                 '-u: end index of query sequences', '-L: start index of reference', '-U: end index of reference',
                 '-d: ref database', '-o: output file', '-O: write mode of output file. w: overwrite, a: append',
                 '-s: spaced seed in comma separated format: 1111,1110,1001',
                 '-r: reduced amino acid alphabet: aa9 (default), aa20, or comma separated groups',
                 '-v: number of hits to show', '-e: expect value', '-m: max ratio of pseudo hits that will trigger stop',
                 '-j: distance between start sites of two neighbor seeds', '-t: filter high frequency kmers whose counts > t',
                 '-F: filter query sequence', '-M: bucket size of hash table', '-c: chunck size of reference',
                 '-a: number of GPUs to use', '-T: tmpdir (accepted, unused)'):
        print('  ' + line)


def parse(argv):
    from .fsearch import parse_flags
    return parse_flags(argv, DEFAULTS)


def resolve(args):
    """-> dict of typed parameters, or None (manual + exit) exactly where the reference bails out."""
    if args['-p'] != 'blastp' or args['-i'] == '' or args['-d'] == '':
        return None
    try:
        p = dict(qry=args['-i'], ref=args['-d'], exp=float(args['-e']), bv=int(args['-v']), start=int(args['-l']),
                 end=int(args['-u']), rstart=int(args['-L']), rend=int(args['-U']), miss=float(args['-m']), thr=int(args['-t']),
                 step=int(args['-j']), flt=args['-F'].upper(), outfile=args['-o'], wrt=args['-O'], ht=int(args['-M']),
                 chk=int(args['-c']), ssd=args['-s'], nr=args['-r'], ngpu=int(args['-a']))
    except ValueError:
        return None
    nr = p['nr'].strip()
    p['nr'] = AA9 if nr == 'aa9' else AA20 if nr == 'aa20' else p['nr']
    p['chk'] = int(p['chk'] / (p['nr'].count('/') + 1))
    return p


def searcher_kwargs(p, device=0):
    return dict(ssd=p['ssd'], nr=p['nr'], ht=p['ht'], chk=p['chk'], step=p['step'], v=p['bv'], thr=p['thr'], expect=p['exp'],
                max_miss=p['miss'], flt=p['flt'], device=device)


def run_single(p):
    from . import fsearch
    s = fsearch.Searcher(**searcher_kwargs(p))
    try:
        s.load_ref(p['ref'], p['rstart'], p['rend'])
        s.load_queries(p['qry'])
        hits = s.search(p['start'], p['end'])
        hits.write(p['outfile'], 'w')
        n = len(hits)
        hits.close()
    finally:
        s.close()
    return n


def run_rank(p):
    """One rank of a multi-GPU run (launched by torch.distributed.run)."""
    import torch
    import torch.distributed as dist
    from . import _lib, dist as sdist, fsearch
    import ctypes as C
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ.get('LOCAL_RANK', 0))
    # SOHIT_BENCH_BACKEND=gloo + SOHIT_BENCH_ONE_GPU=1: functional test of the -a N flow on a 1-GPU box (as in bench.py)
    backend = os.environ.get('SOHIT_BENCH_BACKEND', 'nccl')
    if os.environ.get('SOHIT_BENCH_ONE_GPU'):
        local = 0
    torch.cuda.set_device(local)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    s = fsearch.Searcher(**searcher_kwargs(p, device=local))
    try:
        s.load_ref(p['ref'], p['rstart'], p['rend'])
        s.load_queries(p['qry'])
        lens = s.query_lengths()
        N, D = len(lens), s.num_refs
        st = min(max(0, p['start']), N)
        ed = min(D if p['end'] < 0 else p['end'], N)   # fsearch.py:2980-2981
        lo, hi = sdist.shard_queries(lens, world, st, ed)[rank]
        hits = s.search(lo, hi) if hi > lo else None
        import numpy as np
        g = sdist.gather_records(hits.view_u8() if hits is not None else np.zeros(0, dtype=np.uint8))
        if rank == 0:
            # ranks hold contiguous ascending query ranges: writing their blocks in rank order keeps the file order
            for r, part in enumerate(g.arrays()):
                n = len(part) // C.sizeof(_lib.SoHit)
                arr = (_lib.SoHit * max(n, 1)).from_buffer_copy(part.tobytes() + b'\0' * (C.sizeof(_lib.SoHit) if n == 0 else 0))
                s._chk(s.L.so_write_sc(s.h, arr, n, os.fsencode(p['outfile']), b'w' if r == 0 else b'a'))
        if hits is not None:
            hits.close()
    finally:
        s.close()
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv if argv is None else argv)
    args = parse(argv)
    p = resolve(args)
    if p is None:
        manual_print(os.path.basename(argv[0]) if argv else 'find_hit.py')
        raise SystemExit()
    print('chk size', p['chk'])
    if not p['outfile']:
        print('-o is required (the reference names its part files after it, find_hit.py:110)')
        raise SystemExit()
    if os.path.getsize(p['ref']) >= 4200000000:
        raise SystemExit('reference files >= 4.2e9 bytes need the split/merge path (find_hit.py:303-351), not implemented yet')
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ and int(os.environ['WORLD_SIZE']) > 1:
        run_rank(p)
        return 0
    if p['ngpu'] <= 1:
        run_single(p)
        return 0
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(p['ngpu']), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv[1:]
    env = dict(os.environ)
    env['PYTHONPATH'] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + env.get('PYTHONPATH', '')
    return subprocess.call(cmd, env=env)


if __name__ == '__main__':
    if __package__ in (None, ''):
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from swiftortho_amd.find_hit import main as _m
        sys.exit(_m())
    sys.exit(main())

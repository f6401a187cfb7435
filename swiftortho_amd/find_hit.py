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
The >= 4.2e9-byte reference split + `sort -m | awk` merge (303-351) is reproduced (reference_parts / merge_parts).
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


def query_range(start, end, n_queries, ncpu=1):
    """The query ordinals the reference launcher ends up searching (find_hit.py:95-132): Start < 0 -> 0, End < 0 -> N (the
    QUERY count, not the native's min(D, N) rule of fsearch.py:2981 -- the launcher always passes explicit -l/-u); blocks
    [st, min(N, st + Step)) for st in range(Start, End, Step), Step = max(min(10000, |End - Start| // ncpu), 1).  The last
    block is clipped to N but NOT to End, so up to Step - 1 queries past -u are searched too; reproduced."""
    N = n_queries
    Start = 0 if start < 0 else start
    End = N if end < 0 else end
    Step = max(min(10000, abs(End - Start) // max(1, ncpu)), 1)
    sts = range(Start, End, Step)
    if len(sts) == 0:
        return 0, 0
    lo, hi = min(Start, N), min(N, sts[-1] + Step)
    return lo, max(lo, hi)


def run_single(p, fast_exit=False):
    import time
    if fast_exit:
        os.environ.setdefault('SOHIT_TORCH_PRELOAD', '0')   # this process never imports torch: libsohit binds the system HIP runtime
    laps, t0 = [], time.perf_counter()

    def lap(name):   # SOHIT_TIMING=1: where the wall time of one command goes (stderr)
        nonlocal t0
        t1 = time.perf_counter()
        laps.append((name, t1 - t0))
        t0 = t1
    from . import fsearch
    lap('import')
    s = fsearch.Searcher(**searcher_kwargs(p))
    lap('create')
    try:
        s.load_ref(p['ref'], p['rstart'], p['rend'])
        lap('load_ref')
        s.load_queries(p['qry'])
        lap('load_queries')
        st, ed = query_range(p['start'], p['end'], s.num_queries, p['ngpu'])
        hits = s.search(st, ed)
        lap('index+search')
        hits.write(p['outfile'], 'w')
        lap('write')
        n = len(hits)
        if not fast_exit:
            hits.close()
    finally:
        # (fast_exit still destroys the context: a process that ends with gigabytes of device memory mapped leaves their release to the
        # driver, and the NEXT process's first search waited for it -- 0.2-0.7 s instead of 0.1 in one run of three)
        s.close()
    lap('close')
    if os.environ.get('SOHIT_TIMING'):
        sys.stderr.write('[find_hit] ' + ' '.join('%s=%.3f' % kv for kv in laps) + ' rows=%d\n' % n)
    return n


def run_rank(p):
    """One rank of a multi-GPU run (launched by torch.distributed.run)."""
    import torch
    import torch.distributed as dist
    from . import dist as sdist, fsearch
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
        st, ed = query_range(p['start'], p['end'], len(lens), p['ngpu'])
        # shards balanced by per-query work (index entries visited + residues), not residues alone: family sizes are skewed
        weights = lens.copy()
        if ed > st:
            weights[st:ed] += sdist.sharded_query_work(s, lens, st, ed)   # the pre-pass itself is split over the ranks
        lo, hi = sdist.shard_queries(weights, world, st, ed)[rank]
        # Every rank formats and writes ITS OWN rows (round 5; before: the records of all ranks were gathered to rank 0, which formatted
        # ~300 M rows alone at 1 M proteins).  Ranks hold contiguous ascending query ranges, so the file is the ranks' texts back to back:
        # each rank writes a part file, the byte counts are exchanged (one all_gather of 8 bytes), rank 0 sizes the output file and
        # every rank copies its part to its offset -- find_hit.py's own scheme (block files + cat, find_hit.py:133-146) with the
        # copy done by all ranks at once.  The RCCL record gather stays the exchange of so_search_device / bench.py.
        hits = s.search(lo, hi)   # hi == lo: no rows, still takes part in the exchange
        part = '%s.part%d' % (p['outfile'], rank)
        try:
            hits.write(part, 'w')
            hits.close()
            place_parts(p['outfile'], part, rank, world, dist)
        finally:
            if os.path.exists(part):   # (a rank that failed on the way leaves no part file behind)
                os.remove(part)
    finally:
        s.close()
        dist.destroy_process_group()


def place_parts(outfile, part, rank, world, dist):
    """The ranks' part files -> `outfile`, each copied to the offset its predecessors' sizes give (os.copy_file_range: no user-space
    copy; plain reads / pwrites where the kernel or the file system lacks it); parts are removed.  All ranks must see ONE file system
    (one node: find_hit.py's own part files + `cat` assume the same, find_hit.py:133-146), and the part and the output exist side by
    side for the duration of the copy."""
    import torch
    n = os.path.getsize(part)
    sizes = torch.zeros(world, dtype=torch.int64)
    mine = torch.tensor([n], dtype=torch.int64)
    if dist.get_backend() == 'nccl':
        sizes, mine = sizes.cuda(), mine.cuda()
    dist.all_gather_into_tensor(sizes, mine)
    sizes = [int(x) for x in sizes.cpu().tolist()]
    off = sum(sizes[:rank])
    if rank == 0:
        with open(outfile, 'wb') as f:
            f.truncate(sum(sizes))
    dist.barrier()
    if n:
        with open(part, 'rb') as src, open(outfile, 'r+b') as dst:
            done = 0
            try:
                while done < n:
                    k = os.copy_file_range(src.fileno(), dst.fileno(), min(n - done, 1 << 30), done, off + done)
                    if k <= 0:
                        raise OSError('short copy')
                    done += k
            except (OSError, AttributeError):
                while done < n:
                    buf = os.pread(src.fileno(), min(n - done, 1 << 24), done)
                    if not buf:
                        raise IOError('part file %s shorter than expected' % part)
                    os.pwrite(dst.fileno(), buf, off + done)
                    done += len(buf)
    os.remove(part)
    dist.barrier()
    if rank == 0 and os.path.getsize(outfile) != sum(sizes):
        raise IOError('%s holds %d bytes, the ranks wrote %d' % (outfile, os.path.getsize(outfile), sum(sizes)))


def fasta_parse(f):
    """find_hit.py:23-37 (the launcher's own parser, used only by the split path): header = line minus '>' and its last
    character, sequence lines stripped and joined; a record without sequence lines is dropped."""
    head, seq = '', []
    for i in f:
        if i.startswith('>'):
            if seq:
                yield head, ''.join(seq)
            head, seq = i[1:-1], []
        else:
            seq.append(i.strip())
    if seq:
        yield head, ''.join(seq)


def reference_parts(path, max_chr):
    """find_hit.py:303-344: consecutive records are written to a part until the running character count (header + residues)
    has exceeded max_chr; the record that opens a new part is counted twice (`flag_chr = l_chr` then `+= l_chr`).
    Yields the text of each part ('>%s\\n%s\\n' per record)."""
    cur, flag_chr = [], 0
    with open(path, 'r', encoding='latin-1') as f:
        for hd, sq in fasta_parse(f):
            l_chr = len(hd) + len(sq)
            if flag_chr > max_chr:
                yield ''.join(cur)
                cur, flag_chr = [], l_chr
            flag_chr += l_chr
            cur.append('>%s\n%s\n' % (hd, sq))
    if cur:
        yield ''.join(cur)


def _num_prefix(tok):
    """`sort -n` key: the leading decimal number of the field, 0 when there is none."""
    import re
    m = re.match(rb'\s*(-?\d*\.?\d*)', tok)
    t = m.group(1) if m else b''
    try:
        return float(t) if t not in (b'', b'-', b'.', b'-.') else 0.0
    except ValueError:
        return 0.0


def merge_parts(part_files, bv, out_path):
    """find_hit.py:349: `sort -m -k15,15n -k12,12nr parts/*.sc | awk '{if(c[$1]<bv) print $0;c[$1]+=1}' > OUTFILE`.
    A k-way MERGE (inputs are not re-sorted) of the part files in glob order (bytewise: 0.sc, 1.sc, 10.sc, 2.sc ...) by
    (column 15 numeric, column 12 numeric descending, then the whole line bytewise -- the C-locale last resort; equal lines
    keep file order); then at most bv rows per distinct column 1.  Fields are blank-separated as in sort/awk."""
    import heapq

    def key(line, k):
        f = line.split()
        return (_num_prefix(f[14]) if len(f) > 14 else 0.0, -(_num_prefix(f[11]) if len(f) > 11 else 0.0), line, k)

    files = [open(pf, 'rb') for pf in part_files]
    heap = []
    for k, f in enumerate(files):
        line = f.readline()
        if line:
            heapq.heappush(heap, key(line, k))
    seen = {}
    with open(out_path, 'wb') as out:
        while heap:
            _, _, line, k = heapq.heappop(heap)
            f = line.split()
            q = f[0] if f else b''
            c = seen.get(q, 0)
            if c < bv:
                out.write(line if line.endswith(b'\n') else line + b'\n')
            seen[q] = c + 1
            nxt = files[k].readline()
            if nxt:
                heapq.heappush(heap, key(nxt, k))
    for f in files:
        f.close()


def search_to_file(p, argv, fast_exit=False):
    """One reference file -> one output file: in-process on one GPU, or one rank per GPU through torch.distributed.run."""
    if p['ngpu'] <= 1:
        run_single(p, fast_exit)
        return 0
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:  # a free rendezvous port, so concurrent runs do not collide
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    # -d / -o repeated at the end: the flag grammar keeps the last occurrence (the split path searches part files)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(p['ngpu']), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv[1:] + ['-d', p['ref'], '-o', p['outfile']]
    env = dict(os.environ)
    env.pop('SWIFTORTHO_MAX_CHR', None)  # the ranks search the file they are given
    env['PYTHONPATH'] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + env.get('PYTHONPATH', '')
    return subprocess.call(cmd, env=env)


def main(argv=None, fast_exit=False):
    """fast_exit: the caller ends the process right after (bin/find_hit.py): the one-GPU path skips its own clean-up"""
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
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ and int(os.environ['WORLD_SIZE']) > 1:
        run_rank(p)
        return 0
    # find_hit.py:286-351.  SWIFTORTHO_MAX_CHR stands in for the author's commented test value (line 288).
    max_chr = int(os.environ.get('SWIFTORTHO_MAX_CHR', '4200000000'))
    if os.path.getsize(p['ref']) < max_chr:
        return search_to_file(p, argv, fast_exit)
    # Reference files of >= max_chr bytes: the reference searches consecutive parts as separate databases (so D, the chunk
    # boundaries and the per-chunk thresholds are those of the part) and merges the part outputs.  288 GB of HBM would hold
    # the whole file, but the rows would differ; the observable behaviour is kept instead.
    import shutil
    ref_dir = '%s_parts' % p['ref']
    os.makedirs(ref_dir, exist_ok=True)
    part_ref = os.path.join(ref_dir, 'ref.fsa')
    outs, rc = [], 0
    for k, text in enumerate(reference_parts(p['ref'], max_chr)):
        with open(part_ref, 'w', encoding='latin-1', newline='') as f:
            f.write(text)
        q = dict(p, ref=part_ref, outfile=os.path.join(ref_dir, '%d.sc' % k))
        rc = search_to_file(q, argv) or rc
        outs.append(q['outfile'])
    merge_parts(sorted(o for o in outs if os.path.isfile(o)), p['bv'], p['outfile'])
    shutil.rmtree(ref_dir, ignore_errors=True)
    return rc


if __name__ == '__main__':
    if __package__ in (None, ''):
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from swiftortho_amd.find_hit import main as _m
        sys.exit(_m())
    sys.exit(main())

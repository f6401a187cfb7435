"""Host-side mirror of the reference's native search core interface (lib/fsearch.py), backed
by libsohit.so on an MI355X.

Same names, argument meaning and error behaviour as the reference for this path:

* ``blastp(qry, ref, expect, v, max_miss, st, ed, rst, red, thr, flt, ..., ssd, nr, step, ht, chk)``
  -- fsearch.py:2968 -- generator of hit tuples in the reference's order;
* ``entry_point(argv)`` -- fsearch.py:3152-3264 -- the ``fsearch-c`` flag grammar (``-k v`` or
  ``-kv``, unknown tokens skipped, always returns 0, manual text on bad input) and the
  16-column output row.

The compute is entirely in the HIP library; this module never falls back to a CPU path.
"""
import ctypes as C
import os
import sys

from . import _lib


class _LazyNumpy:
    """numpy is imported by the first function that uses it: the one-GPU command line (find_hit.py -> Searcher -> Hits.write) never does,
    and its import is ~0.1 s of a 0.7 s command."""

    def __getattr__(self, name):
        import numpy
        globals()["np"] = numpy
        return getattr(numpy, name)


np = _LazyNumpy()

AA9 = "AST,CFILMVY,DN,EQ,G,H,KR,P,W"
AA20 = "A,S,T,C,F,I,L,M,V,Y,D,N,E,Q,G,H,K,R,P,W"


class SohitError(RuntimeError):
    pass


class Searcher:
    """One libsohit context == one GPU (include/sohit.h).  Not thread-safe."""

    def __init__(self, ssd="111111", nr=AA9, ht=120000000, chk=50000, step=1, v=500, thr=-1, expect=1e-3, max_miss=1e-3,
                 flt="T", device=0, profile=False):
        self.L = _lib.load()
        self._keep = (ssd.encode(), nr.encode())
        p = _lib.SoParams(self._keep[0], self._keep[1], int(ht), int(chk), int(step), int(v), int(thr), float(expect),
                          float(max_miss), 1 if flt == "T" else 0, 1 if profile else 0)
        self.h = self.L.so_create(int(device), C.byref(p))
        if not self.h:
            raise SohitError(self.L.so_last_error(None).decode())
        self.nc = int(self.L.so_bucket_count(self.h))  # -M, or the reference's `bins` default when -M < 1

    def close(self):
        if getattr(self, "h", None):
            self.L.so_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc:
            raise SohitError(self.L.so_last_error(self.h).decode())

    # reference side ---------------------------------------------------------------------------
    def load_ref(self, path, rst=-1, red=-1):
        self._chk(self.L.so_load_ref(self.h, os.fsencode(path), rst, red))

    def load_ref_bytes(self, data, rst=-1, red=-1):
        self._chk(self.L.so_load_ref_mem(self.h, data, len(data), rst, red))

    def build_index(self):
        self._chk(self.L.so_build_index(self.h))

    def drop_index(self):
        self._chk(self.L.so_drop_index(self.h))

    def load_index(self, prefix):
        """Fasta.load (fsearch.py:2355-2444): make the chunk indexes `<prefix>.<k>.idx/.soas/.bin` resident instead of building them
        (the reference FASTA must be loaded: the files hold no sequences)."""
        self._chk(self.L.so_load_index(self.h, os.fsencode(prefix)))

    # query side -------------------------------------------------------------------------------
    def load_queries(self, path):
        self._chk(self.L.so_load_queries(self.h, os.fsencode(path)))

    def load_queries_bytes(self, data):
        self._chk(self.L.so_load_queries_mem(self.h, data, len(data)))

    @property
    def num_queries(self):
        return int(self.L.so_num_queries(self.h))

    @property
    def num_refs(self):
        return int(self.L.so_num_refs(self.h))

    def query_lengths(self):
        n = self.num_queries
        return np.array([self.L.so_query_len(self.h, i) for i in range(n)], dtype=np.int64)

    def ref_lengths(self):
        n = self.num_refs
        return np.array([self.L.so_ref_len(self.h, i) for i in range(n)], dtype=np.int64)

    # search -----------------------------------------------------------------------------------
    def search(self, st=-1, ed=-1):
        """Queries [st, ed) of the loaded query file -> Hits (owning wrapper)."""
        hits = C.POINTER(_lib.SoHit)()
        n = C.c_int64(0)
        self._chk(self.L.so_search_loaded(self.h, st, ed, C.byref(hits), C.byref(n)))
        return Hits(self, hits, n.value)

    def search_device(self, st=-1, ed=-1):
        """Like search(), but the so_hit records stay in HBM (multi-GPU path: exchanged over RCCL before any host copy)."""
        n = C.c_int64(0)
        ptr = C.c_void_p()
        self._chk(self.L.so_search_device(self.h, st, ed, C.byref(ptr), C.byref(n)))
        return DeviceHits(self, n.value)

    def query_work(self, st=-1, ed=-1):
        """Index entries each query of [st, ed) visits over all chunks (pre-pass, no search): the shard-balancing weight."""
        N = self.num_queries
        lo = min(max(0, st), N)
        hi = min(N if ed < 0 else ed, N)
        out = np.zeros(max(hi - lo, 1), dtype=np.uint64)
        self._chk(self.L.so_query_work(self.h, lo, hi, out.ctypes.data))
        return out[:max(hi - lo, 0)].astype(np.int64)

    def counters(self):
        c = _lib.SoCounters()
        self.L.so_get_counters(self.h, C.byref(c))
        return c.as_dict()

    def set_profile(self, on):
        self.L.so_set_profile(self.h, 1 if on else 0)

    def set_option(self, name, value):
        """one switch of csrc/tune.h by its SOHIT_* name, for this searcher from now on (so_create read the environment once)"""
        if self.L.so_set_option(self.h, str(name).encode(), str(value).encode()) != 0:
            raise ValueError(self.L.so_last_error(self.h).decode())

    def reset_counters(self):
        self.L.so_reset_counters(self.h)

    def timing(self):
        n = self.L.so_timing_report(self.h, None, 0)
        buf = C.create_string_buffer(n + 2)
        self.L.so_timing_report(self.h, buf, n + 1)
        return {kv.split("=")[0]: float(kv.split("=")[1]) for kv in buf.value.decode().split(";") if kv}

    # introspection (tests) ----------------------------------------------------------------------
    def chunk_threshold(self, k):
        return int(self.L.so_chunk_threshold(self.h, k))

    def chunk_index(self, k):
        n = int(self.L.so_chunk_entries(self.h, k))
        start = np.zeros(self.nc + 1, dtype=np.uint32)
        ent = np.zeros(max(n, 1), dtype=np.uint64)
        self._chk(self.L.so_chunk_download(self.h, k, start.ctypes.data, ent.ctypes.data))
        return start, ent[:n]

    def masked_query(self, q):
        n = self.L.so_masked_query(self.h, q, None, 0)
        if n < 0:
            return None
        buf = C.create_string_buffer(max(1, n))
        self.L.so_masked_query(self.h, q, buf, n)
        return buf.raw[:n]

    def query_candidates(self, q):
        n = self.L.so_query_candidates(self.h, q, None, 0)
        if n < 0:
            return None
        out = np.zeros((max(n, 1), 4), dtype=np.uint32)
        self.L.so_query_candidates(self.h, q, out.ctypes.data, n)
        return out[:n]


class DeviceHits:
    """so_hit records of one so_search_device() call, resident in the ctx's device memory until its next search."""
    record_bytes = C.sizeof(_lib.SoHit)

    def __init__(self, s, n):
        self.s, self.n = s, n

    def __len__(self):
        return self.n

    def tensor(self, device=None):
        """uint8 torch tensor (n * 80 bytes) on the ctx's GPU holding a copy of the records (device-to-device)."""
        import torch
        t = torch.empty(self.n * self.record_bytes, dtype=torch.uint8, device=device or torch.device("cuda", torch.cuda.current_device()))
        # the block comes from torch's caching allocator on torch's stream and is filled from the library's own stream: nothing torch
        # queued earlier on that block may still be running when the copy starts
        torch.cuda.current_stream(t.device).synchronize()
        self.s._chk(self.s.L.so_device_hits_copy(self.s.h, C.c_void_p(t.data_ptr()), self.n))
        return t

    def close(self):
        pass


def hits_from_bytes(s, data):
    """so_hit records gathered from other ranks (bytes / uint8 array) -> ctypes array usable with so_write_sc / so_format_hit.
    A writable contiguous uint8 array (the pinned staging buffer of the gather) is wrapped in place: a 1 M-protein result is 24 GB
    and must not be copied again on rank 0."""
    rec = C.sizeof(_lib.SoHit)
    if isinstance(data, np.ndarray) and data.dtype == np.uint8 and data.flags.c_contiguous and data.flags.writeable and data.size >= rec:
        n = data.size // rec
        arr = (_lib.SoHit * n).from_buffer(data)
        return arr, n
    buf = np.ascontiguousarray(data, dtype=np.uint8).tobytes()
    n = len(buf) // rec
    arr = (_lib.SoHit * max(n, 1)).from_buffer_copy(buf + b"\0" * (rec if n == 0 else 0))
    return arr, n


class Hits:
    """Result rows of one search; frees the library buffer on close."""

    def __init__(self, s, ptr, n):
        self.s, self.ptr, self.n = s, ptr, n

    def __len__(self):
        return self.n

    def array(self):
        """numpy structured view (copy) of the so_hit records."""
        dt = np.dtype([(n, t) for n, t in (("qidx", "<i8"), ("sidx", "<i8"), ("identity", "<f8"), ("evalue", "<f8"), ("aln", "<i4"),
                                          ("mis", "<i4"), ("gap", "<i4"), ("qst", "<i4"), ("qed", "<i4"), ("sst", "<i4"),
                                          ("sed", "<i4"), ("bit", "<i4"), ("qlen", "<i4"), ("slen", "<i4"), ("matches", "<i4"),
                                          ("ungapped", "<i4"))])
        if self.n == 0:
            return np.zeros(0, dtype=dt)
        nbytes = self.n * C.sizeof(_lib.SoHit)  # can exceed 2 GiB (1M x 1M: 24 GB): no string_at
        view = (C.c_char * nbytes).from_address(C.addressof(self.ptr.contents))
        return np.frombuffer(view, dtype=dt).copy()

    def write(self, path, mode="w"):
        self.s._chk(self.s.L.so_write_sc(self.s.h, self.ptr, self.n, os.fsencode(path), mode.encode()))

    def rows(self):
        buf = C.create_string_buffer(1 << 16)
        out = []
        for i in range(self.n):
            k = self.s.L.so_format_hit(self.s.h, C.byref(self.ptr[i]), buf, len(buf))
            if k < 0:
                raise SohitError(self.s.L.so_last_error(self.s.h).decode())
            if k >= len(buf):  # so_format_hit returns the size it needs: very long headers
                buf = C.create_string_buffer(int(k) + 1)
                k = self.s.L.so_format_hit(self.s.h, C.byref(self.ptr[i]), buf, len(buf))
            out.append(buf.raw[:k])
        return out

    def view_u8(self):
        """uint8 numpy view of the record buffer (no copy; valid until close())."""
        if not self.n:
            return np.zeros(0, dtype=np.uint8)
        nbytes = self.n * C.sizeof(_lib.SoHit)
        return np.frombuffer((C.c_char * nbytes).from_address(C.addressof(self.ptr.contents)), dtype=np.uint8)

    def raw_bytes(self):
        if not self.n:
            return b""
        nbytes = self.n * C.sizeof(_lib.SoHit)
        return bytes((C.c_char * nbytes).from_address(C.addressof(self.ptr.contents)))

    def close(self):
        if self.ptr:
            self.s.L.so_free_hits(self.ptr)
            self.ptr = None

    __del__ = close


def blastp(qry, ref, expect=1e-5, v=500, max_miss=1e-3, st=-1, ed=-1, rst=-1, red=-1, thr=-1, flt="T", ref_idx="", memory=True,
           ssd="", nr="", step=4, ht=-1, chk=100000, tmpdir="./tmpdir", device=0):
    """fsearch.py:2968 -- yields (i, j, li, lj, idy, aln, mis, gap, qst, qed, sst, sed, e, bit, seed) per reported row
    (query ordinal, subject ordinal, lengths, identity, ..., 1-based starts as printed, e-value, bit, ungapped score)."""
    s = Searcher(ssd=ssd, nr=nr, ht=ht, chk=chk, step=step, v=v, thr=thr, expect=expect, max_miss=max_miss, flt=flt, device=device)
    try:
        s.load_ref(ref, rst, red)
        s.load_queries(qry)
        hits = s.search(st, ed)
        for r in hits.array():
            yield (int(r["qidx"]), int(r["sidx"]), int(r["qlen"]), int(r["slen"]), float(r["identity"]), int(r["aln"]), int(r["mis"]),
                   int(r["gap"]), int(r["qst"]), int(r["qed"]), int(r["sst"]), int(r["sed"]), float(r["evalue"]), int(r["bit"]),
                   int(r["ungapped"]))
        hits.close()
    finally:
        s.close()


def makedb(ref, space='11111111', nr=AA9, step=1, ht=-1, chk=500000, device=0):
    """fsearch.py:2809-2814 + Fasta.makedb / Fasta.write (2283-2352): build the chunk indexes of `ref` (on the GPU) and write them
    in the reference's on-disk format, `<ref>.<k>.idx` (locus: int32 per index entry, a bucket's entries in the reference's
    slot order = descending insertion order), `<ref>.<k>.soas` (prefix lengths of the chunk's sequences) and `<ref>.<k>.bin`
    (start[NC] + the trailer `offset;offend;max weight;threshold;NC;seeds;alphabet` + one length byte).  Nothing in the
    reference's own entry point reads these files back (`entry_point` accepts `-p blastp` only and `blastp` always indexes in
    memory); they are written for tools that consume the format.  Returns the list of (start, end) chunk ranges."""
    s = Searcher(ssd=space, nr=nr, ht=ht, chk=chk, step=step, device=device)
    out = []
    try:
        s.load_ref(ref)
        s.build_index()
        lens = s.ref_lengths()
        N = len(lens)
        mw = max(sp.count('1') for sp in space.split(','))
        NC = s.nc
        for k, i in enumerate(range(0, N, chk)):
            st, ed = i, min(i + chk, N)
            start, ent = s.chunk_index(k)
            soas = np.concatenate([[0], np.cumsum(lens[st:ed])]).astype(np.int64)
            # (so_chunk_download hands the entries over in the reference's slot order: bucket ascending, entry descending inside a bucket --
            # the device puts them in that order, order_chunk in host_index.hip; until round 6 a host lexsort of the chunk did)
            assert int(start[NC]) == len(ent)
            locus = soas[(ent >> np.uint64(32)).astype(np.int64)] + (ent & np.uint64(0xFFFFFF)).astype(np.int64)
            name = '%s.%d' % (ref, i // chk)
            locus.astype('<i4').tofile(name + '.idx')
            soas.astype('<i4').tofile(name + '.soas')
            trailer = '%d;%d;%d;%d;%d;%s;%s' % (st, ed + 1, mw, s.chunk_threshold(k), NC, space, nr)
            with open(name + '.bin', 'wb') as f:
                start[:NC].astype('<i4').tofile(f)
                f.write(trailer.encode('latin-1') + bytes([len(trailer) & 0xFF]))
            out.append((st, ed))
    finally:
        s.close()
    return out


def index_params(name):
    """The parameter trailer of one chunk's `.bin` file as Fasta.load reads it (fsearch.py:2380-2387): the last byte is the length
    of `offset;offend;max weight;threshold;NC;seeds;alphabet`."""
    with open(name + '.bin', 'rb') as f:
        f.seek(0, os.SEEK_END)
        n = f.tell()
        f.seek(n - 1)
        m = f.read(1)[0]
        start = max(n - m - 1, 0)
        f.seek(start)
        para = f.read(m).decode('latin-1')
    offset, offend, mw, thr, nc, space, nr = para.split(';')
    return dict(offset=int(offset), offend=int(offend), mw=int(mw), threshold=int(thr), NC=int(nc), space=space, nr=nr)


def load(ref, name=None, device=0, **search_kw):
    """Fasta.load (fsearch.py:2355-2444) for every chunk `makedb` wrote: a Searcher over `ref` whose index comes from the files
    `<name>.<k>.idx/.soas/.bin` (name defaults to `ref`, as makedb names them), created with the seeds, alphabet and bucket count
    of the files' own trailer; `search_kw` = the search-side parameters (v, expect, max_miss, thr, flt, step is irrelevant here)."""
    name = name or ref
    p = index_params('%s.0' % name)
    s = Searcher(ssd=p['space'], nr=p['nr'], ht=p['NC'], device=device, **search_kw)
    try:
        s.load_ref(ref)
        s.load_index(name)
    except Exception:
        s.close()
        raise
    return s


def manual_print(out=None):
    w = (out or sys.stdout).write
    w("Usage:\n  fsearch -p blastp -i qry.fsa -d db.fsa\nParameters:\n")
    for line in ("-p: program", "-i: query sequences in fasta format", "-l: start index of query sequences",
                 "-u: end index of query sequences", "-L: start index of reference", "-U: end index of reference",
                 "-d: ref database", "-o: output file", "-O: write mode of output file. w: overwrite, a: append",
                 "-s: spaced seed in format: 1111,1110,1001.. etc",
                 "-r: reduced amino acid alphabet in format: AST,CFILMVY,DN,EQ,G,H,KR,P,W", "-v: number of hits to show",
                 "-e: expect value", "-m: max ratio of pseudo hits that will trigger stop",
                 "-j: distance between start sites of two neighbor seeds, greater will reduce the size of database",
                 "-t: filter high frequency kmers whose counts > t", "-F: Filter query sequence",
                 "-M: bucket size of hash table", "-c: chunck size of reference. default is 50K",
                 "-T: tmpdir (accepted, unused: nothing is spilled to disk)"):
        w("  %s\n" % line)


DEFAULTS = {'-p': '', '-v': '500', '-s': '111111', '-i': '', '-d': '', '-e': '1e-3', '-l': '-1', '-u': '-1', '-m': '1e-3',
            '-t': '-1', '-r': AA9, '-j': '4', '-F': 'T', '-o': '', '-D': '', '-O': 'wb', '-L': '-1', '-U': '-1', '-M': '-1',
            '-c': '50000', '-T': './tmpdir'}


def parse_flags(argv, defaults):
    """fsearch.py:3189-3199: `-k v` or `-kv`; unknown tokens are skipped."""
    args = dict(defaults)
    n = len(argv)
    for i in range(1, n):
        k = argv[i]
        if k in args:
            if i + 1 < n:
                args[k] = argv[i + 1]
        elif k[:2] in args and len(k) > 2:
            args[k[:2]] = k[2:]
    return args


def entry_point(argv, device=0, searcher=None):
    """fsearch.py:3152 -- same flags as lib/fsearch-c; writes the 16-column rows; always returns 0."""
    args = parse_flags(argv, DEFAULTS)
    if args['-p'] != 'blastp' or args['-i'] == '' or args['-d'] == '':
        manual_print()
        return 0
    try:
        exp, bv, start, end = float(args['-e']), int(args['-v']), int(args['-l']), int(args['-u'])
        rstart, rend, miss, thr = int(args['-L']), int(args['-U']), float(args['-m']), int(args['-t'])
        step, ht, chk = int(args['-j']), int(args['-M']), int(args['-c'])
    except ValueError:
        print('blastp')
        manual_print()
        return 0
    wrt = args['-O']
    wrt = wrt if (wrt and wrt in 'wa') else 'w'
    try:
        s = searcher or Searcher(ssd=args['-s'], nr=args['-r'], ht=ht, chk=chk, step=step, v=bv, thr=thr, expect=exp, max_miss=miss,
                                 flt=args['-F'], device=device)
    except SohitError as e:  # e.g. -M < 1 with an odd seed weight (the reference crashes there); the native always returns 0
        sys.stderr.write('fsearch: %s\n' % e)
        return 0
    try:
        s.load_ref(args['-d'], rstart, rend)
        s.load_queries(args['-i'])
        hits = s.search(start, end)
        if args['-o']:
            hits.write(args['-o'], wrt)
        else:
            for r in hits.rows():
                sys.stdout.write(r.decode('latin-1'))
        hits.close()
    finally:
        if searcher is None:
            s.close()
    return 0


if __name__ == "__main__":
    sys.exit(entry_point(sys.argv))

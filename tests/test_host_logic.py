"""CPU-only tests of the Python host layer: flag grammar, find_hit defaults/blocking, synthetic
proteome determinism, query sharding, and the 2-rank gloo gather (the N > 1 path)."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_fsearch_flag_grammar():
    from swiftortho_amd import fsearch
    a = fsearch.parse_flags(["fsearch", "-p", "blastp", "-i", "q.fa", "-dref.fa", "-e1e-5", "bogus", "-v", "10", "--x"], fsearch.DEFAULTS)
    assert a["-p"] == "blastp" and a["-i"] == "q.fa" and a["-d"] == "ref.fa" and a["-e"] == "1e-5" and a["-v"] == "10"
    assert a["-j"] == "4" and a["-M"] == "-1" and a["-c"] == "50000"      # fsearch-c defaults (fsearch.py:3187-3188)
    # a trailing flag without a value keeps its default (the reference would raise IndexError there)
    assert fsearch.parse_flags(["x", "-v"], fsearch.DEFAULTS)["-v"] == "500"


def test_entry_point_prints_manual_and_returns_zero(capsys):
    from swiftortho_amd import fsearch
    assert fsearch.entry_point(["fsearch"]) == 0
    assert "Usage" in capsys.readouterr().out
    assert fsearch.entry_point(["fsearch", "-p", "blastp", "-i", "x"]) == 0      # -d missing
    assert fsearch.entry_point(["fsearch", "-p", "blastp", "-i", "x", "-d", "y", "-v", "abc"]) == 0   # bad int -> manual


def test_find_hit_defaults_and_chunk_rule():
    from swiftortho_amd import find_hit
    a = find_hit.parse(["find_hit.py", "-p", "blastp", "-i", "q", "-d", "r", "-o", "out", "-r", "aa20", "-a", "4"])
    p = find_hit.resolve(a)
    assert p["ssd"] == "11111111" and p["ht"] == 120000000 and p["step"] == 1 and p["bv"] == 500 and p["exp"] == 1e-3
    assert p["nr"] == find_hit.AA20 and p["chk"] == 50000 and p["ngpu"] == 4
    two = find_hit.resolve(find_hit.parse(["x", "-p", "blastp", "-i", "q", "-d", "r", "-r", find_hit.AA9 + "/" + find_hit.AA20]))
    assert two["chk"] == 25000                                  # chk = int(-c / number of alphabets), find_hit.py:273-274
    assert find_hit.resolve(find_hit.parse(["x", "-p", "makedb", "-i", "q"])) is None
    assert find_hit.resolve(find_hit.parse(["x", "-p", "blastp", "-i", "q"])) is None
    assert find_hit.resolve(find_hit.parse(["x", "-p", "blastp", "-i", "q", "-d", "r", "-e", "zz"])) is None


def test_find_hit_cli_manual_exit():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "find_hit.py")], capture_output=True, text=True)
    assert "Usage" in r.stdout and r.returncode == 0


def test_synthprot_is_deterministic():
    from swiftortho_amd import synthprot
    a, b = synthprot.synthprot(300, 200, 9), synthprot.synthprot(300, 200, 9)
    assert a == b and a.count(b">") == 300
    assert hashlib.md5(synthprot.synthprot(10000, 300)).hexdigest() == "311b5d33882ea0ded2a82dd88084bf4d"
    u = synthprot.uniform_proteins(50, 100, 1)
    assert u.count(b">") == 50 and all(len(l) == 100 for l in u.split(b"\n")[1::2])


def test_shard_queries_balanced_contiguous():
    from swiftortho_amd.dist import shard_queries
    rng = np.random.default_rng(0)
    lens = rng.integers(50, 2000, 1000)
    for world in (1, 2, 3, 8):
        sh = shard_queries(lens, world)
        assert sh[0][0] == 0 and sh[-1][1] == 1000
        assert all(sh[i][1] == sh[i + 1][0] for i in range(world - 1))
        tot = [int(lens[a:b].sum()) for a, b in sh]
        assert max(tot) - min(tot) <= 2 * 2000
    assert shard_queries(lens, 4, 100, 200)[0][0] == 100 and shard_queries(lens, 4, 100, 200)[-1][1] == 200
    assert shard_queries([], 3) == [(0, 0)] * 3
    assert shard_queries([5], 4)[-1][1] == 1


def test_shards_balanced_by_work_on_skewed_families():
    """Work per query (seed hits, extensions, alignments) grows with the size of its family; families are skewed.  Shards cut
    by the per-query work estimate keep max / mean rank work <= 1.1 where residue-balanced shards do not."""
    from swiftortho_amd.dist import imbalance, shard_queries
    rng = np.random.default_rng(7)
    fam = rng.zipf(1.6, 20000).clip(1, 400)           # family size of each query's family, taxon-major order mixes them
    lens = rng.integers(100, 600, fam.size)
    work = fam.astype(np.int64) * fam * 300 + lens    # ~ members x hits per member
    for world in (2, 4, 8):
        by_work = shard_queries(work, world)
        assert by_work[0][0] == 0 and by_work[-1][1] == fam.size and all(by_work[i][1] == by_work[i + 1][0] for i in range(world - 1))
        assert imbalance(work, by_work) <= 1.1
    assert imbalance(work, shard_queries(lens, 8)) > imbalance(work, shard_queries(work, 8))
    # one dominant query cannot be split: it gets a shard of its own and the rest is still cut sensibly
    w = np.ones(1000, dtype=np.int64)
    w[500] = 10000
    sh = shard_queries(w, 4)
    assert any(a <= 500 < b and b - a <= 260 for a, b in sh)


def test_launcher_query_range_rule():
    """find_hit.py:97-116: End < 0 -> the QUERY count; the last block is clipped to N, not to -u"""
    from swiftortho_amd.find_hit import query_range
    assert query_range(-1, -1, 70, 1) == (0, 70)
    assert query_range(5, 48, 70, 3) == (5, 61)             # Step = 43 // 3 = 14: blocks 5, 19, 33, 47 -> [47, 61)
    assert query_range(0, 15000, 100000, 1) == (0, 20000)   # Step = 10000
    assert query_range(0, 15000, 12000, 1) == (0, 12000)
    assert query_range(10, 5, 70, 1) == (0, 0)
    assert query_range(80, -1, 70, 1) == (0, 0)
    assert query_range(0, 7, 70, 16) == (0, 7)              # Step = max(7 // 16, 1) = 1


WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch.distributed as dist
from swiftortho_amd import dist as sdist
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
lens = np.arange(1, 101)
lo, hi = sdist.shard_queries(lens, world)[rank]
# fake 80-byte hit records: one per query of the shard, tagged with the query ordinal
recs = np.zeros((hi - lo, 10), dtype=np.int64); recs[:, 0] = np.arange(lo, hi)
if rank == 1: recs = recs[:0] if os.environ.get("EMPTY1") else recs
parts = sdist.gather_bytes(recs.tobytes())
# the work pre-pass split over the ranks: a stand-in searcher whose "work" of query i is 3 i + 1
class S:
    def query_work(self, a, b): return np.arange(a, b, dtype=np.int64) * 3 + 1
w = sdist.sharded_query_work(S(), lens, 10, 90)
assert np.array_equal(w, np.arange(10, 90) * 3 + 1), w[:5]
assert len(sdist.sharded_query_work(S(), lens, 40, 40)) == 0
if rank == 0:
    allr = np.frombuffer(b"".join(parts), dtype=np.int64).reshape(-1, 10)
    print("GATHERED", len(allr), int(allr[:, 0].sum()), bool(np.all(np.diff(allr[:, 0]) > 0)))
else:
    assert parts is None
dist.destroy_process_group()
'''


@pytest.mark.parametrize("empty1", [False, True])
def test_two_rank_gloo_gather(tmp_path, empty1):
    """world_size-2 run of the sharding + gatherv used by bench.py / find_hit.py -a N (gloo on CPU)."""
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300 + (7 if empty1 else 0)), WORLD_SIZE="2")
    if empty1:
        env["EMPTY1"] = "1"
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0][0].splitlines() if l.startswith("GATHERED")][0].split()
    if empty1:
        lo1 = None
        assert line[3] == "True" and int(line[1]) < 100
    else:
        assert line[1:] == ["100", str(sum(range(100))), "True"]


PLACE_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch.distributed as dist
from swiftortho_amd import find_hit
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
out = os.environ["OUT"]
part = "%%s.part%%d" %% (out, rank)
# rank r's "rows": r * 1000 + 17 lines (rank 1 none when EMPTY1), each naming its rank and line number
n = 0 if (rank == 1 and os.environ.get("EMPTY1")) else rank * 1000 + 17
open(part, "wb").write(b"".join(b"rank%%d line %%07d\n" %% (rank, i) for i in range(n)))
find_hit.place_parts(out, part, rank, world, dist)
assert not os.path.exists(part)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,empty1", [(2, False), (3, True)])
def test_every_rank_places_its_own_rows_in_the_output_file(tmp_path, world, empty1):
    """find_hit.py -a N (round 5): every rank writes a part file and copies it to ITS offset of the output file (one all_gather of the
    byte counts; no rank formats or writes another's rows).  world_size 2 and 3 over gloo, one rank without rows: the file is the
    ranks' texts back to back, the parts are gone."""
    script = tmp_path / "pw.py"
    script.write_text(PLACE_WORKER % ROOT)
    out = tmp_path / "out.sc"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29950 + os.getpid() % 40 + world), WORLD_SIZE=str(world), OUT=str(out))
    if empty1:
        env["EMPTY1"] = "1"
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    want = b"".join(b"rank%d line %07d\n" % (r, i) for r in range(world) for i in range(0 if (r == 1 and empty1) else r * 1000 + 17))
    assert out.read_bytes() == want
    assert sorted(os.listdir(tmp_path)) == ["out.sc", "pw.py"]


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torch.distributed environment starts the two ranks itself (child torchrun, before
    anything touches the GPU) and relays the exit code.  Here there is no GPU, so both ranks must fail loudly -- "needs a GPU:
    libsohit has no CPU path" -- and the launcher must report failure; the same command on the GPU box is in test_gpu_parity.py."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "c2",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode != 0
    # both ranks were started (the launcher's failure report names rank 1) and at least the first one to fail said why: the
    # launcher terminates the other rank as soon as one has failed, sometimes before it has printed its own message
    assert p.stderr.count("bench.py needs a GPU") >= 1, p.stderr[-3000:]
    assert "local_rank: 1" in p.stderr, p.stderr[-3000:]
    assert '"metric"' not in p.stdout


def test_row_formatter_equals_printf_and_oracle_f2s(oracle):
    """The library formats "%f" itself (exact 128-bit integer arithmetic, round-half-even like glibc) and builds f2s (fsearch.py:43-61)
    on it: identical to Python's '%f' and to the oracle's f2s over identities, mantissas, near-integers, exact ties (odd / 128),
    e-values of every magnitude, negative values and the snprintf fallbacks (huge, inf, nan)."""
    import ctypes as C
    import numpy as np
    from swiftortho_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(5)
    aln = rng.integers(1, 5000, 200000)
    parts = [rng.integers(0, aln + 1) * (100. / aln),                       # identities
             10 ** rng.random(100000), -rng.random(50000) * 330, np.round(-rng.random(50000) * 330) + rng.normal(0, 1e-9, 50000),
             (2 * rng.integers(0, 10 ** 6, 50000) + 1) / 128.0, -(2 * rng.integers(0, 10 ** 4, 5000) + 1) / 128.0,   # exact ties
             10 ** rng.uniform(-320, 3, 200000), 10 ** rng.uniform(-6, 1, 50000), rng.random(50000) * 1e-6, rng.random(20000) * 1e14,
             np.array([0.0, -0.0, 1e-3, 9.9999e-4, 1e-5, 2.88e-261, 0.5, 1e15, 1e16, 1e300, np.inf, -np.inf, np.nan, 5e-324, 999999.9999995, 0.9999995, 9.9999995]),
             rng.integers(0, 2 ** 63, size=100000, dtype=np.uint64).view(np.float64)]
    v = np.ascontiguousarray(np.concatenate(parts), dtype=np.float64)
    out = np.empty(len(v) * 1400, dtype=np.uint8)
    w = L.so_fmt_rows(C.c_void_p(v.ctypes.data), len(v), C.c_void_p(out.ctypes.data), len(out))
    assert w > 0
    rows = out[:w].tobytes().decode().split("\n")[:-1]
    assert len(rows) == len(v)
    for x, row in zip(v.tolist(), rows):
        f, e = row.split("\t")
        assert f == "%f" % x, (x, f)
        assert e == oracle.f2s(x), (x, e, oracle.f2s(x))


def test_frequency_cap_keeps_every_window_when_their_total_fits():
    """What k_cap_all (csrc/k_seed.hip) relies on to skip the k-mer orders: the reference's cap walks a query's windows in score order and
    tests the running total BEFORE adding the window's bucket size (fsearch.py:2667-2677), so when all bucket sizes together stay at or
    below the limit no window is ever refused -- in any order.  Above the limit the kept set does depend on the order."""
    rng = np.random.default_rng(5)

    def capped(counts, order, limit):   # the reference's loop: `if total > limit: break; total += n; keep`
        kept, total = [], 0
        for p in order:
            if total > limit:
                break
            total += int(counts[p])
            kept.append(int(p))
        return sorted(kept)

    order_matters = 0
    for _ in range(300):
        n = int(rng.integers(1, 60))
        counts = rng.integers(0, 40, size=n)
        limit = int(rng.integers(0, 1200))
        a, b = capped(counts, rng.permutation(n), limit), capped(counts, rng.permutation(n), limit)
        if int(counts.sum()) <= limit:
            assert a == b == list(range(n))
        else:
            order_matters += a != b
    assert order_matters > 0

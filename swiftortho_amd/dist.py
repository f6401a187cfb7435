"""Multi-GPU plumbing for the search: one process per GPU, queries sharded across ranks, the
reference index replicated (each rank builds it locally: deterministic, no broadcast), and the
per-rank hit records gathered to rank 0 with torch.distributed (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference's equivalent is find_hit.py's pool of
`fsearch-c` processes over query blocks joined with `cat` (find_hit.py:107-146).

Shards are contiguous query ranges (so the concatenation of per-rank outputs is in ascending query
order, like the reference's block files) balanced by a per-query WEIGHT: the number of index entries
the query visits (Searcher.query_work(): one bounds + cap pre-pass) plus its residues -- seed hits,
ungapped extensions and alignments all grow with the size of the query's family, residues do not.
The exchange is a size-exact gatherv of records that are still in HBM: one all_gather of byte counts,
then ONE collective with uneven splits (all_to_all_single: every rank sends its records to rank 0 and
nothing to the others -- RCCL runs it as one group of point-to-point transfers, which is what a gatherv
is on xGMI's point-to-point links) into ONE destination buffer of sum(sizes) bytes on rank 0 (no padding
to the largest rank), and a device-to-host copy there in pinned slabs.  SOHIT_GATHER=p2p selects the
hand-written isend / irecv batch of rounds 2-3 instead (same bytes, same buffer).
"""
import numpy as np


def shard_queries(weights, world, lo=0, hi=None):
    """Contiguous query ranges [lo_r, hi_r), r = 0..world-1, balanced by `weights` (per-query cost; residues or work).
    Contiguity keeps the concatenation of per-rank outputs in ascending query order."""
    lengths = np.asarray(weights, dtype=np.int64)
    hi = len(lengths) if hi is None or hi < 0 else min(hi, len(lengths))
    lo = max(0, lo)
    if hi <= lo:
        return [(lo, lo)] * world
    cum = np.concatenate([[0], np.cumsum(lengths[lo:hi])])
    total = cum[-1]
    bounds = [lo]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        # the boundary nearest to the target (a huge query next to the cut goes to the lighter side)
        if k > 0 and k < len(cum) and abs(cum[k - 1] - target) <= abs(cum[k] - target):
            k -= 1
        bounds.append(lo + min(max(k, bounds[-1] - lo), hi - lo))
    bounds.append(hi)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def imbalance(weights, shards):
    """max / mean of the per-shard weight sums"""
    w = np.asarray(weights, dtype=np.int64)
    sums = np.array([int(w[a:b].sum()) for a, b in shards], dtype=np.float64)
    return float(sums.max() / max(sums.mean(), 1e-300))


class GatheredDevice:
    """Result of gather_device_records on the destination rank: `sizes` (bytes per rank) and ONE tensor of sum(sizes)
    bytes holding the ranks' records back to back in rank order (on the GPU with RCCL, on the host with gloo)."""

    def __init__(self, buf, sizes):
        self.buf, self.sizes = buf, sizes
        self._host = None

    def to_host(self):
        """-> uint8 numpy array of all records.  Up to 1 GiB: one copy into a reusable pinned buffer, returned as is; larger results (a
        1 M-protein search gathers 24 GB) go through two pinned 256 MiB slabs into ordinary memory -- pinning tens of GB is slow
        and can fail."""
        if self._host is None:
            if self.buf.is_cuda:
                import torch
                n = self.buf.numel()
                if n <= SLAB_LIMIT:
                    host = _staging("recv", n, True)
                    host[:n].copy_(self.buf, non_blocking=True)
                    torch.cuda.current_stream().synchronize()
                    self._host = host[:n].numpy()
                else:
                    out = np.empty(n, dtype=np.uint8)
                    slabs = [_staging("slab%d" % k, SLAB_BYTES, True) for k in (0, 1)]
                    evs = [torch.cuda.Event(), torch.cuda.Event()]
                    pend = [None, None]   # (offset, bytes) in flight per slab
                    k = 0
                    for off in range(0, n, SLAB_BYTES):
                        m = min(SLAB_BYTES, n - off)
                        if pend[k]:
                            evs[k].synchronize()
                            o, mm = pend[k]
                            out[o:o + mm] = slabs[k][:mm].numpy()
                        slabs[k][:m].copy_(self.buf[off:off + m], non_blocking=True)
                        evs[k].record()
                        pend[k] = (off, m)
                        k ^= 1
                    for j in (k, k ^ 1):
                        if pend[j]:
                            evs[j].synchronize()
                            o, mm = pend[j]
                            out[o:o + mm] = slabs[j][:mm].numpy()
                    self._host = out
            else:
                self._host = self.buf.numpy()
        return self._host

    def arrays(self):
        h = self.to_host()
        offs = np.concatenate([[0], np.cumsum(self.sizes)])
        return [h[offs[r]:offs[r + 1]] for r in range(len(self.sizes))]


SLAB_LIMIT = 1 << 30
SLAB_BYTES = 1 << 28


def gather_device_records(t, dst=0):
    """Gather one uint8 torch tensor of packed hit records per rank to rank `dst`, size-exact.

    With the RCCL backend `t` is a device tensor and nothing touches the host: all_gather of the byte counts, then ONE
    all_to_all_single with uneven splits (a rank's input goes to `dst` whole, `dst` receives every rank's bytes into disjoint
    slices of a single sum(sizes)-byte device buffer).  A collective every rank enters with the communicator the all_gather
    has already used: no point-to-point pair is set up lazily on the first exchange.  SOHIT_GATHER=p2p: the same transfers as
    an isend / irecv batch.  With gloo (CPU tests / one-GPU functional runs) the same flow runs on host tensors."""
    import os
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = dist.get_backend() == "nccl"
    t = t.reshape(-1)
    if not on_gpu and t.is_cuda:
        t = t.cpu()
    dev = t.device
    n = int(t.numel())
    sizes_t = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes_t, torch.tensor([n], dtype=torch.int64, device=dev))
    sizes = [int(x) for x in sizes_t.cpu().tolist()]
    if os.environ.get("SOHIT_GATHER", "alltoall") != "p2p":
        total = sum(sizes) if rank == dst else 0
        recv = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)[:total]
        dist.all_to_all_single(recv, t, output_split_sizes=sizes if rank == dst else [0] * world,
                               input_split_sizes=[n if r == dst else 0 for r in range(world)])
        return GatheredDevice(recv, sizes) if rank == dst else None
    if rank == dst:
        total = sum(sizes)
        recv = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)[:total]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        if n:
            recv[offs[rank]:offs[rank + 1]].copy_(t)
        ops = [dist.P2POp(dist.irecv, recv[offs[r]:offs[r + 1]], r) for r in range(world) if r != dst and sizes[r]]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return GatheredDevice(recv, sizes)
    if n:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, dst)]):
            w.wait()
    return None


def allgather_ragged(arr):
    """int64 numpy array per rank (any lengths) -> their concatenation in rank order, on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    on_gpu = dist.get_backend() == "nccl"
    dev = "cuda" if on_gpu else "cpu"
    a = np.ascontiguousarray(arr, dtype=np.int64)
    sizes_t = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes_t, torch.tensor([len(a)], dtype=torch.int64, device=dev))
    sizes = [int(x) for x in sizes_t.cpu().tolist()]
    m = max(max(sizes), 1)
    mine = torch.zeros(m, dtype=torch.int64, device=dev)
    if len(a):
        mine[:len(a)] = torch.from_numpy(a).to(dev)
    allt = torch.empty(world * m, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allt, mine)
    h = allt.cpu().numpy()
    return np.concatenate([h[r * m:r * m + sizes[r]] for r in range(world)]) if sum(sizes) else np.zeros(0, dtype=np.int64)


def sharded_query_work(searcher, lens, st, ed):
    """Per-query seed-hit work of queries [st, ed) (Searcher.query_work) with the pre-pass itself split over the ranks: every rank
    hashes / bounds / caps a residue-balanced share of the queries, one small all_gather puts the shares together.  (Until round 4
    every rank ran the pre-pass over ALL queries before the shards existed.)"""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_queries(lens, world, st, ed)[rank]
    mine = searcher.query_work(lo, hi) if hi > lo else np.zeros(0, dtype=np.int64)
    return allgather_ragged(mine)


_pinned = {}  # reusable pinned staging tensors, keyed by role


def _staging(role, nbytes, pin):
    import torch
    buf = _pinned.get(role)
    if buf is None or buf.numel() < nbytes:
        # head room for the next, slightly larger result -- but not on multi-GB gathers (a 1 M-protein result is 24 GB of records)
        cap = max(nbytes, 1 << 16)
        buf = torch.empty(cap * 5 // 4 if cap < (1 << 30) else cap, dtype=torch.uint8, pin_memory=pin)
        _pinned[role] = buf
    return buf


def gather_bytes(payload, dst=0):
    """bytes in, list of bytes out on `dst` (host-side convenience over gather_device_records; tests)."""
    import torch
    import torch.distributed as dist
    t = torch.frombuffer(bytearray(payload), dtype=torch.uint8) if payload else torch.zeros(0, dtype=torch.uint8)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    g = gather_device_records(t, dst)
    return None if g is None else [p.tobytes() for p in g.arrays()]

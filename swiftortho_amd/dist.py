"""Multi-GPU plumbing for the search: one process per GPU, queries sharded across ranks, the
reference index replicated (each rank builds it locally: deterministic, no broadcast), and the
per-rank hit records gathered to rank 0 with torch.distributed (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference's equivalent is find_hit.py's pool of
`fsearch-c` processes over query blocks joined with `cat` (find_hit.py:107-146).
"""
import numpy as np


def shard_queries(lengths, world, lo=0, hi=None):
    """Contiguous query ranges [lo_r, hi_r), r = 0..world-1, balanced by residues.
    Contiguity keeps the concatenation of per-rank outputs in ascending query order."""
    lengths = np.asarray(lengths, dtype=np.int64)
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
        bounds.append(lo + min(max(k, bounds[-1] - lo), hi - lo))
    bounds.append(hi)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


_pinned = {}  # reusable pinned staging tensors, keyed by role


def _staging(role, nbytes, pin):
    import torch
    buf = _pinned.get(role)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 16) * 5 // 4, dtype=torch.uint8, pin_memory=pin)
        _pinned[role] = buf
    return buf


class Gathered:
    """Result of gather_records on the destination rank: `sizes` (bytes per rank) is available at once, the
    records themselves after the device-to-host copy has finished (`arrays()` waits for it), so a caller that
    pipelines -- bench.py -- lets that copy run under its next step."""

    def __init__(self, host, sizes, mx, event):
        self._host, self.sizes, self._mx, self._event = host, sizes, mx, event

    def arrays(self):
        if self._event is not None:
            self._event.synchronize()
        h = self._host.numpy()
        return [h[r * self._mx:r * self._mx + n] for r, n in enumerate(self.sizes)]


_send_done = [None]  # event after the last host-to-device copy out of the pinned send buffer


def gather_records(view, device=None, dst=0):
    """Gather one uint8 numpy array (the packed hit records of this rank) per rank to rank `dst`.

    Returns a `Gathered` on `dst` (its arrays are views into one pinned host buffer, valid until the next
    call), None elsewhere.  One all_gather of the sizes, one padded gather (gatherv) over RCCL, one
    device-to-host copy; staging buffers are pinned and reused."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = dist.get_backend() == "nccl"
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    view = np.ascontiguousarray(view, dtype=np.uint8).reshape(-1)
    n = int(view.nbytes)
    sizes_t = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes_t, torch.tensor([n], dtype=torch.int64, device=device))
    sizes = [int(x) for x in sizes_t.cpu().tolist()]
    mx = max(max(sizes), 1)
    send = _staging("send", mx, on_gpu)
    if on_gpu and _send_done[0] is not None:
        _send_done[0].synchronize()  # the previous call's copy out of this buffer
    if n:
        send.numpy()[:n] = view
    if on_gpu:
        gsend = send[:mx].to(device, non_blocking=True)
        _send_done[0] = torch.cuda.Event()
        _send_done[0].record()
    else:
        gsend = send[:mx]
    if rank == dst:
        recv = torch.empty(world * mx, dtype=torch.uint8, device=device)
        dist.gather(gsend, [recv[r * mx:(r + 1) * mx] for r in range(world)], dst=dst)
        event = None
        if on_gpu:
            host = _staging("recv", world * mx, True)
            host[:world * mx].copy_(recv, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
        else:
            host = recv
        return Gathered(host, sizes, mx, event)
    dist.gather(gsend, None, dst=dst)
    return None


def gather_bytes(payload, device=None, dst=0):
    """bytes in, list of bytes out on `dst` (copies; the bench and the CLI use gather_records)."""
    g = gather_records(np.frombuffer(payload, dtype=np.uint8) if payload else np.zeros(0, dtype=np.uint8), device, dst)
    return None if g is None else [p.tobytes() for p in g.arrays()]

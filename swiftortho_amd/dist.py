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


def gather_bytes(payload, device=None, dst=0):
    """Gather one bytes object per rank to rank `dst` (others get None).

    all_gather of the sizes, then a padded gather of the payload (gatherv).  Hit records are tens
    of bytes per reported row, so this is latency- not bandwidth-bound."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = torch.zeros(mx, dtype=torch.uint8, device=device)
    if payload:
        buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    if rank == dst:
        out = [torch.zeros(mx, dtype=torch.uint8, device=device) for _ in range(world)]
        dist.gather(buf, out, dst=dst)
        return [out[r][:sizes[r]].cpu().numpy().tobytes() for r in range(world)]
    dist.gather(buf, None, dst=dst)
    return None

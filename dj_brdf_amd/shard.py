"""How the path shards across GPUs: independent units, zero exchange.

* eval / pdf / sample: pair-index ranges, one contiguous block per rank (the BRDF object is
  replicated: <= 23 MB);
* batch fit: materials dealt round-robin, one rank per GPU.
No data-path collective exists; ``gather_rows`` is only the host-side concatenation of the
per-rank result rows (what params.txt needs) and works on any torch.distributed backend.
"""
from __future__ import annotations


def block_range(n: int, world: int, rank: int):
    """[lo, hi) of rank's contiguous block when n items are split as evenly as possible."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def round_robin(n: int, world: int, rank: int):
    """indices rank owns when n independent units are dealt round-robin."""
    return list(range(rank, n, world))


def gather_rows(rows, world: int, rank: int, n: int):
    """Reassemble per-rank [(index, payload)] lists into input order on every rank.

    Uses torch.distributed.all_gather_object (a control-plane gather of a few bytes per
    material; not a data-path collective)."""
    import torch.distributed as dist
    if world == 1 or not dist.is_initialized():
        merged = list(rows)
    else:
        parts = [None] * world
        dist.all_gather_object(parts, list(rows))
        merged = [r for part in parts for r in part]
    out = [None] * n
    for idx, payload in merged:
        out[idx] = payload
    return out

"""Chromosome-bucket sharding of a query batch over the GPUs of one node.

The reference already buckets regions by seqid before querying (commands/intersect.rs:114-120);
queries are independent and the index is tiny, so the index is replicated on every GPU and the
*queries* are partitioned: whole chromosome buckets are placed by LPT (largest first onto the
least-loaded rank) and a bucket that would overshoot the ideal load is split -- the remainder goes
back into the pool.  No data-path collective is needed; the only exchange is the all-gather of
per-rank hit counts (``allgather_hit_counts``).
"""
from __future__ import annotations

import heapq
from typing import List, Sequence, Tuple

import numpy as np

Slice = Tuple[int, int, int]  # (chr, lo, hi): rows lo..hi of that chromosome's bucket (BED order)


def plan_shards(bucket_sizes: Sequence[int], n_ranks: int, tolerance: float = 0.02) -> List[List[Slice]]:
    """Deterministic plan: for every rank the list of (chr, lo, hi) bucket slices it owns."""
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    total = int(sum(int(x) for x in bucket_sizes))
    plan: List[List[Slice]] = [[] for _ in range(n_ranks)]
    if total == 0:
        return plan
    ideal = -(-total // n_ranks)
    slack = max(1, int(ideal * tolerance))
    # max-heap of pending pieces (size, chr, lo, hi); ties broken by chr for determinism
    pend = [(-int(sz), c, 0, int(sz)) for c, sz in enumerate(bucket_sizes) if int(sz) > 0]
    heapq.heapify(pend)
    loads = [(0, r) for r in range(n_ranks)]
    heapq.heapify(loads)
    while pend:
        negsz, c, lo, hi = heapq.heappop(pend)
        sz = -negsz
        load, r = heapq.heappop(loads)
        room = ideal - load
        if sz > room + slack and room > slack:
            # split: fill this rank up to the ideal, return the rest to the pool
            plan[r].append((c, lo, lo + room))
            heapq.heappush(pend, (-(sz - room), c, lo + room, hi))
            heapq.heappush(loads, (load + room, r))
        else:
            plan[r].append((c, lo, hi))
            heapq.heappush(loads, (load + sz, r))
    for p in plan:
        p.sort()
    return plan


def bucket_regions(regions: np.ndarray, n_chr: int):
    """Stable bucketing by chr (BED order kept inside a bucket, like intersect.rs:114-120).

    Returns (order, bucket_offsets): ``order`` permutes the rows into bucket order."""
    chr_ = regions[:, 0].astype(np.int64)
    if len(chr_) and int(chr_.max()) >= n_chr:
        raise IndexError("region chr %d out of range (%d seqids)" % (int(chr_.max()), n_chr))
    order = np.argsort(chr_, kind="stable")
    sizes = np.bincount(chr_, minlength=n_chr)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    return order, offs


def shard_rows(regions: np.ndarray, n_chr: int, n_ranks: int, rank: int) -> np.ndarray:
    """Row indices (into ``regions``) owned by ``rank`` under the plan for this batch."""
    order, offs = bucket_regions(regions, n_chr)
    plan = plan_shards(np.diff(offs), n_ranks)
    parts = [order[offs[c] + lo: offs[c] + hi] for c, lo, hi in plan[rank]]
    return np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)


def allgather_hit_counts(n_queries: int, n_hits: int, device=None):
    """The path's one exchange step: every rank learns every rank's (n_queries, n_hits).

    Uses torch.distributed (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" on CPU).
    Messages are 16 bytes per rank, i.e. latency-bound: it is issued once per batch."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return np.array([[n_queries, n_hits]], dtype=np.int64)
    mine = torch.tensor([int(n_queries), int(n_hits)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return torch.stack(out).cpu().numpy()


def allgather_root_bitmap(words: np.ndarray, device=None) -> np.ndarray:
    """The CLI's exchange step: every rank learns the union of the ranks' root-hit bitmaps.

    ``words`` = this rank's packed bitmap (u64 words over the index's roots in sorted order, as
    QueryBatch.root_bitmap packs it).  RCCL has no bitwise-OR reduction, and max/sum over packed words is not an OR
    once a chromosome bucket is split across ranks, so the packed bitmaps (<= 8 KB each at 63 k roots) are
    all-gathered and OR-ed locally.  Returns the OR-ed words."""
    import torch
    import torch.distributed as dist

    w = np.ascontiguousarray(words, dtype=np.uint64)
    if not (dist.is_available() and dist.is_initialized()):
        return w.copy()
    mine = torch.from_numpy(w.view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return np.bitwise_or.reduce(torch.stack(out).cpu().numpy().view(np.uint64), axis=0)


def allgather_seqid_hits(per_seqid_hits: np.ndarray, device=None) -> np.ndarray:
    """Per-seqid kept-pair counts of every rank ((world, n_seq) i64; a few hundred bytes per rank)."""
    import torch
    import torch.distributed as dist

    h = np.ascontiguousarray(per_seqid_hits, dtype=np.int64)
    if not (dist.is_available() and dist.is_initialized()):
        return h[None, :].copy()
    mine = torch.from_numpy(h.copy())
    if device is not None:
        mine = mine.to(device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return torch.stack(out).cpu().numpy()

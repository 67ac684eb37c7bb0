"""Multi-GPU sharding of a batch: harts are independent systems, so a batch
of N harts is cut into contiguous hart-index ranges, one per rank (one process per GPU), with no
exchange while running.  The only collective is the final gather of the per-hart result records
(status, counts, digests) plus a max/sum reduction of the timing; RCCL on the GPUs ("nccl"
backend), gloo in the CPU tests.  Nothing here touches HIP: tensors are moved to `device` only
because the nccl backend needs device tensors.
"""
from __future__ import annotations

import torch


def partition(n_total: int, world: int) -> list[tuple[int, int]]:
    """Contiguous ranges [g*N/G, (g+1)*N/G) for g in 0..G-1 (strong split of one batch)."""
    return [(g * n_total // world, (g + 1) * n_total // world) for g in range(world)]


def weak_range(per_rank: int, rank: int) -> tuple[int, int]:
    """Weak scaling (bench.py): every rank owns `per_rank` harts, global ids rank*per_rank..."""
    return rank * per_rank, (rank + 1) * per_rank


def gather_results(raw: bytes, dist=None, device=None) -> list[bytes]:
    """All-gather the packed wsp_result records of every rank.  Ranks may hold different numbers of records (a strong split of a
    list that does not divide evenly): lengths are exchanged first and the payloads padded to the longest.  Returns the list of
    byte strings in rank (= hart) order; world size 1 or dist None returns [raw]."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [bytes(raw)]
    world = dist.get_world_size()
    n = torch.tensor([len(raw)], dtype=torch.int64)
    if device is not None:
        n = n.to(device)
    lens = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(lens, n)
    lens = [int(t.item()) for t in lens]
    longest = max(lens)
    rec = torch.frombuffer(bytearray(raw) + bytearray(longest - len(raw)), dtype=torch.uint8) if longest else torch.empty(0, dtype=torch.uint8)
    if device is not None:
        rec = rec.to(device)
    out = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(out, rec)
    return [bytes(t.cpu().numpy().tobytes())[:k] for t, k in zip(out, lens)]


def reduce_totals(seconds: float, units: float, dist=None, device=None) -> tuple[float, float]:
    """(max over ranks of seconds, sum over ranks of units)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds), float(units)
    t = torch.tensor([seconds], dtype=torch.float64)
    u = torch.tensor([units], dtype=torch.float64)
    if device is not None:
        t = t.to(device); u = u.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())

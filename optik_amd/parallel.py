"""Multi-GPU partition of the restart stream (one process per GPU, RCCL over xGMI).

Restarts are independent and their seeds are a pure function of the global restart
index (lib.rs:358-370), so rank g of G takes a contiguous index range and no data
moves between GPUs while solving.  The only exchange is the selection of
lib.rs:397-413 across ranks: a min-all-reduce of the 8-byte key (Speed: the index
itself; Quality: the bit pattern of ||x - x0||_2, which is order-preserving for
non-negative doubles), then of the index among the ranks holding the minimum key.
"""
from __future__ import annotations

import torch

I64_MAX = torch.iinfo(torch.int64).max


def shard_range(begin: int, end: int, rank: int, world: int):
    """Contiguous split of [begin, end) (lower ranks get the remainder)."""
    total = end - begin
    base, rem = divmod(total, world)
    lo = begin + rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def local_key(bufs, mode: str):
    """(key, idx) int64 tensors [T] of this rank's winners; I64_MAX where none."""
    idx = bufs["win_idx"]  # uint64 viewed as int64: UINT64_MAX == -1
    none = idx < 0
    if mode == "speed":
        key = idx.clone()
    else:
        key = bufs["win_key"].view(torch.int64).clone()  # non-negative f64 bits sort like the value
    key[none] = I64_MAX
    idx = torch.where(none, torch.full_like(idx, I64_MAX), idx)
    return key, idx


def select_winner(bufs, mode: str, distributed: bool):
    """Global winner index per target (int64 [T], I64_MAX = no solution anywhere).
    Two tiny RCCL min-all-reduces; with one rank it is the local result."""
    key, idx = local_key(bufs, mode)
    if not distributed:
        return idx
    import torch.distributed as dist
    gkey = key.clone()
    dist.all_reduce(gkey, op=dist.ReduceOp.MIN)
    cand = torch.where(key == gkey, idx, torch.full_like(idx, I64_MAX))
    dist.all_reduce(cand, op=dist.ReduceOp.MIN)
    return cand


def gather_winner_x(bufs, winner_idx, begin: int, end: int, distributed: bool):
    """x and f of the global winner on every rank: the owning rank contributes its row,
    the others zeros, and a sum-all-reduce (<= 64 B per target) broadcasts it."""
    x = bufs["win_x"].clone()
    f = bufs["win_f"].clone()
    if not distributed:
        return x, f
    import torch.distributed as dist
    mine = (winner_idx >= begin) & (winner_idx < end) & (winner_idx == bufs["win_idx"])
    x = torch.where(mine[:, None], x, torch.zeros_like(x))
    f = torch.where(mine, f, torch.zeros_like(f))
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return x, f

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


def _all_reduce(t, op):
    """dist.all_reduce in place; with the gloo backend (CPU tests, single-GPU smoke runs of the
    multi-rank path) device tensors are staged through the host."""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def select_winner(bufs, mode: str, distributed: bool):
    """Global winner index per target (int64 [T], I64_MAX = no solution anywhere).
    Two tiny RCCL min-all-reduces; with one rank it is the local result."""
    key, idx = local_key(bufs, mode)
    if not distributed:
        return idx
    import torch.distributed as dist
    gkey = key.clone()
    _all_reduce(gkey, dist.ReduceOp.MIN)
    cand = torch.where(key == gkey, idx, torch.full_like(idx, I64_MAX))
    _all_reduce(cand, dist.ReduceOp.MIN)
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
    _all_reduce(x, dist.ReduceOp.SUM)
    _all_reduce(f, dist.ReduceOp.SUM)
    return x, f

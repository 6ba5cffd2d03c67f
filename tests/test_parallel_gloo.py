"""World-size-2 CPU tests (gloo) of the multi-GPU selection logic: restart-range sharding
and the two min-all-reduces that pick the global winner (optik_amd/parallel.py).  The
kernels are not involved: each rank fabricates the per-target winner records a launch
over its own restart range would leave in HBM."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from optik_amd.parallel import I64_MAX, gather_winner_x, select_winner, shard_range


def test_shard_range_partitions_exactly():
    for total, world in [(65536 * 8, 8), (1000, 3), (5, 8), (4194304, 8)]:
        pieces = [shard_range(0, total, r, world) for r in range(world)]
        assert pieces[0][0] == 0 and pieces[-1][1] == total
        for a, b in zip(pieces, pieces[1:]):
            assert a[1] == b[0]
        sizes = [hi - lo for lo, hi in pieces]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        T, n = 5, 7
        begin, end = shard_range(0, 2000, rank, world)
        # target 0: both ranks have a success; 1: only rank 1; 2: nobody; 3: equal keys (tie ->
        # lower index); 4: only rank 0
        idx = torch.tensor([[40, -1, -1, 10, 7], [1500, 1200, -1, 1800, -1]][rank], dtype=torch.int64)
        key = torch.tensor([[0.9, 0.0, 0.0, 0.5, 0.3], [0.2, 0.7, 0.0, 0.5, 0.0]][rank], dtype=torch.float64)
        bufs = dict(win_idx=idx, win_key=key,
                    win_x=torch.full((T, n), float(rank + 1), dtype=torch.float64),
                    win_f=torch.full((T,), 0.5 * (rank + 1), dtype=torch.float64))
        w = select_winner(bufs, mode, True)
        x, f = gather_winner_x(bufs, w, begin, end, True)
        q.put((rank, w.tolist(), x[:, 0].tolist(), f.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["speed", "quality"])
def test_two_rank_winner_selection(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if mode == "speed":   # lowest successful index wins (lib.rs:409-412, 1-thread order)
        want = [40, 1200, I64_MAX, 10, 7]
        owner = [1, 2, 0, 1, 1]
    else:                 # min ||x - x0||, ties to the lower index (lib.rs:398-408)
        want = [1500, 1200, I64_MAX, 10, 7]
        owner = [2, 2, 0, 1, 1]
    for rank, w, x0col, f in res:
        assert w == want
        assert x0col == [float(o) for o in owner]   # the owning rank's x reaches every rank
        assert f == [0.5 * o for o in owner]


# ---- world size 8: the width the driver's scaling sweep ends at (lib.rs:397-413 over the ranks) ----

R8 = 2003          # restarts per target: NOT divisible by 8 -- shards of 251 and 250
T8 = 4096          # config 5's targets, cut 8 ways


def _worker8(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 7
        begin, end = shard_range(0, R8, rank, world)
        none = -1
        # what a launch over [begin, end) would leave per target (index -1 = no solution on this rank):
        #  0  the SAME Quality key on ranks 1, 4 and 6 (smaller than everyone else's): the lowest index (rank 1's) wins;
        #     Speed: rank 0's index is the lowest
        #  1  a solution on rank 5 only -- nothing on 7 of 8 ranks
        #  2  nobody
        #  3  every rank solved its FIRST index (the shard boundaries, ragged): Speed -> 0; Quality -> rank 7 has the smallest key
        #  4  every rank solved its LAST index, all keys equal: a tie across all eight -> the lowest index
        #  5  ranks 2 and 3 only, rank 3 closer to the seed
        idx = [begin + 3, none, none, begin, end - 1, none]
        key = [0.5 + 0.01 * rank, 0.0, 0.0, 1.0 - 0.1 * rank, 0.125, 0.0]
        if rank in (1, 4, 6):
            key[0] = 0.25
        if rank == 5:
            idx[1], key[1] = begin + 17, 0.75
        if rank in (2, 3):
            idx[5], key[5] = begin + 5, (0.9 if rank == 2 else 0.4)
        T = len(idx)
        bufs = dict(win_idx=torch.tensor(idx, dtype=torch.int64), win_key=torch.tensor(key, dtype=torch.float64),
                    win_x=torch.full((T, n), float(rank + 1), dtype=torch.float64),
                    win_f=torch.full((T,), 0.5 * (rank + 1), dtype=torch.float64))
        w = select_winner(bufs, mode, True)
        x, f = gather_winner_x(bufs, w, begin, end, True)
        # config 5: the targets are cut per rank, no collective on the data path -- every rank reports its part
        parts = [None] * world
        dist.all_gather_object(parts, shard_range(0, T8, rank, world))
        q.put((rank, w.tolist(), x[:, 0].tolist(), f.tolist(), (begin, end), parts))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["speed", "quality"])
def test_eight_rank_winner_selection(mode):
    """Eight ranks: ragged restart shards, a three-way tie on the Quality key, a solution on one rank of eight, none at
    all, a tie across all eight, and config 5's 4 096 targets cut eight ways."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    shards = [r[4] for r in res]
    assert shards[0][0] == 0 and shards[-1][1] == R8 and all(a[1] == b[0] for a, b in zip(shards, shards[1:]))
    assert sorted({hi - lo for lo, hi in shards}) == [250, 251]
    b = [s[0] for s in shards]
    e = [s[1] for s in shards]
    if mode == "speed":   # the lowest solved index (lib.rs:409-412 in its deterministic reading)
        want = [b[0] + 3, b[5] + 17, I64_MAX, b[0], e[0] - 1, b[2] + 5]
        owner = [1, 6, 0, 1, 1, 3]
    else:                 # the smallest ||x - x0||, ties to the lower index (lib.rs:398-408)
        want = [b[1] + 3, b[5] + 17, I64_MAX, b[7], e[0] - 1, b[3] + 5]
        owner = [2, 6, 0, 8, 1, 4]
    for rank, w, x0col, f, _, parts in res:
        assert w == want, (rank, w, want)
        assert x0col == [float(o) for o in owner]      # the owning rank's x reaches every rank, once
        assert f == [0.5 * o for o in owner]
        assert parts == [(512 * r, 512 * (r + 1)) for r in range(world)]

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

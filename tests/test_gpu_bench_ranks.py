"""bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU): on the
one-GPU test box both ranks share cuda:0 and talk gloo (OPTIK_BENCH_ONE_DEVICE / _BACKEND -- the
real runs use nccl = RCCL and one GPU per rank).  Two ranks of R restarts per step must report the
winners one rank finds in 2 R restarts: the restart ranges are sharded, the winner is the global
minimum."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    rows = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    assert len(rows) == 1, out[-2000:]
    return json.loads(rows[0])


@pytest.mark.gpu
def test_two_ranks_report_the_single_rank_winners():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    common = ["--steps", "4", "--warmup", "1", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, "bench.py", "--restarts", "4096", *common], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    env = dict(os.environ, OPTIK_BENCH_BACKEND="gloo", OPTIK_BENCH_ONE_DEVICE="1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
                          "--restarts", "2048", *common], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = _line(one.stdout), _line(two.stdout)
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["scaling"] == "weak"
    assert b["config"]["winner_index_per_step"] == a["config"]["winner_index_per_step"]
    assert b["value"] > 0 and b["roofline"]["frac"] > 0

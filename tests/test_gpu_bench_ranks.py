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


def _bench(extra, ranks, port=None):
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", *extra]
    if ranks == 1:
        r = subprocess.run([sys.executable, "bench.py", *common], cwd=ROOT, capture_output=True, text=True, timeout=600)
    else:
        env = dict(os.environ, OPTIK_BENCH_BACKEND="gloo", OPTIK_BENCH_ONE_DEVICE="1")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                            "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", str(ranks),
                            *common], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return _line(r.stdout)


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.gpu
def test_config4_quality_strong_scaling_two_ranks():
    """BASELINE.json config 4 in small: SolutionMode::Quality, one restart range cut into a
    contiguous part per rank, min-all-reduce of ||x - x0|| then of the index."""
    extra = ["--mode", "quality", "--scaling", "strong", "--restarts", "8192"]
    a, b = _bench(extra, 1), _bench(extra, 2, _port())
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and b["config"]["solution_mode"] == "quality"
    assert b["config"]["restarts_per_gpu"] == 4096 and a["config"]["restarts_per_gpu"] == 8192
    assert b["config"]["winner_index_per_step"] == a["config"]["winner_index_per_step"]
    assert all(w > 0 for w in a["config"]["winner_index_per_step"])


@pytest.mark.gpu
def test_config5_targets_sharded_two_ranks():
    """BASELINE.json config 5 in small: independent targets cut into one part per rank, no
    collective; every target is solved by its own rank exactly as by a single rank."""
    extra = ["--targets", "64", "--restarts", "128"]
    a, b = _bench(extra, 1), _bench(extra, 2, _port())
    assert a["unit"] == b["unit"] == "ik calls/s"
    assert b["config"]["parallelism"] == "targets x2"
    # rank 0 reports its own targets: the first 32 of every step
    assert b["config"]["winner_index_per_step"] == a["config"]["winner_index_per_step"]
    assert a["config"]["solved_targets"] >= 3 * 60 and b["config"]["solved_targets"] >= 3 * 30


@pytest.mark.gpu
def test_plain_gpus_flag_spawns_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it must run TWO ranks (it re-executes under
    torch.distributed.run) and say so: n_gpus == 2, both ranks listed, the 1-rank winners."""
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--reps", "2"]
    one = subprocess.run([sys.executable, "bench.py", "--restarts", "4096", *common], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(OPTIK_BENCH_BACKEND="gloo", OPTIK_BENCH_ONE_DEVICE="1")
    two = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--restarts", "2048", *common], cwd=ROOT,
                         env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = _line(one.stdout), _line(two.stdout)
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2
    assert b["config"]["world"] == 2 and b["config"]["backend"] == "gloo"
    assert sorted(d["rank"] for d in b["config"]["rank_devices"]) == [0, 1]
    assert len({d["pid"] for d in b["config"]["rank_devices"]}) == 2
    assert b["config"]["winner_index_per_step"] == a["config"]["winner_index_per_step"]
    assert len(b["config"]["value_reps"]) == 2 and b["config"]["value_min"] <= b["value"] <= b["config"]["value_max"]
    # first-contact evidence of a multi-rank run: every rank's device identity, the latency of each collective
    assert all("uuid" in d and "name" in d for d in b["config"]["rank_devices"])
    cu = b["config"]["collective_us"]
    assert set(cu) == {"all_reduce_min_int64_8B", "all_reduce_sum_f64_64B", "barrier"} and all(v > 0 for v in cu.values())
    assert a["config"]["collective_us"] is None


@pytest.mark.gpu
def test_headline_command_on_two_ranks_also_times_configs_4_and_5():
    """The driver's scaling sweep runs ONE command per N: with N > 1 the headline line carries config 4 (Quality, strong)
    and config 5 (targets cut per rank) as well -- here two ranks sharing the box's GPU over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(OPTIK_BENCH_BACKEND="gloo", OPTIK_BENCH_ONE_DEVICE="1")
    two = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--reps", "2",
                          "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    b = _line(two.stdout)
    oc = b["config"]["other_configs"]
    assert b["n_gpus"] == 2 and set(oc) == {"config4_strong_quality", "config5_targets"}
    assert oc["config4_strong_quality"]["restarts_per_gpu"] == (1 << 22) // 2 and oc["config4_strong_quality"]["restarts_per_s"] > 0
    assert all(w >= 0 for w in oc["config4_strong_quality"]["winner_index_per_step"])
    assert oc["config5_targets"]["targets_per_gpu"] == 2048 and oc["config5_targets"]["ik_calls_per_s"] > 0


@pytest.mark.gpu
def test_headline_command_on_eight_ranks_gives_a_well_formed_line():
    """The width the driver's scaling sweep ends at: `bench.py --gpus 8` (two steps), eight ranks sharing the test box's
    GPU over gloo.  The line must be well formed: world 8, eight distinct processes, the eight restart ranges of the
    weak-scaling headline, config 4 cut into 524 288 restarts per rank and config 5 into 512 targets per rank
    (VERDICT r5 item 6)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(OPTIK_BENCH_BACKEND="gloo", OPTIK_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--reps", "1",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    b = _line(r.stdout)
    assert b["n_gpus"] == 8 and b["config"]["world"] == 8 and b["scaling"] == "weak" and b["steps"] == 2
    assert sorted(d["rank"] for d in b["config"]["rank_devices"]) == list(range(8))
    assert len({d["pid"] for d in b["config"]["rank_devices"]}) == 8
    assert b["config"]["restarts_per_gpu"] == 65536 and b["config"]["parallelism"] == "restart-range x8"
    assert b["value"] > 0 and b["ms_per_step"] > 0 and len(b["config"]["winner_index_per_step"]) == 2
    oc = b["config"]["other_configs"]
    assert oc["config4_strong_quality"]["restarts_per_gpu"] == (1 << 22) // 8
    assert oc["config5_targets"]["targets_per_gpu"] == 512
    assert set(b["config"]["collective_us"]) == {"all_reduce_min_int64_8B", "all_reduce_sum_f64_64B", "barrier"}


@pytest.mark.gpu
def test_inprocess_two_devices_reports_two_gpus():
    """`--inprocess --gpus 2`: one process, two device contexts (the test box's GPU listed twice)
    behind optik_robot_set_devices; the winner is the one a single device finds in the same range.  n_gpus is what
    the calls were really cut over (optik_robot_last_parts): a range too short to be worth cutting stays on one."""
    common = ["--inprocess", "--steps", "2", "--warmup", "1", "--reps", "1", "--restarts", "16384"]
    env = dict(os.environ, OPTIK_BENCH_ONE_DEVICE="1")
    two = subprocess.run([sys.executable, "bench.py", "--gpus", "2", *common], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    common[-1] = "32768"
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1", *common], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    a, b = _line(one.stdout), _line(two.stdout)
    assert b["n_gpus"] == 2 and b["config"]["inprocess"] and b["config"]["devices"] == [0, 0]
    assert b["config"]["parts_per_call"] == [2] and a["n_gpus"] == 1
    assert a["config"]["winner_index_per_step"] == b["config"]["winner_index_per_step"]
    common[-1] = "2048"  # 4096 restarts over two devices: not worth cutting, and the line says so
    small = subprocess.run([sys.executable, "bench.py", "--gpus", "2", *common], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=600)
    assert small.returncode == 0, small.stderr[-2000:]
    c = _line(small.stdout)
    assert c["n_gpus"] == 1 and c["config"]["devices_configured"] == 2


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["plain", "torchrun"])
def test_one_rank_goes_through_rccl(launcher):
    """The process-group path on the real backend: ONE rank, backend nccl (= RCCL on ROCm) --
    init_process_group(device_id=...), all_gather_object, the int64 MIN / f64 SUM / f64 MAX all-reduces on
    device tensors and the barriers all execute on the GPU, and the winners are those of the plain run."""
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--reps", "2", "--restarts", "4096"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "OPTIK_BENCH_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    plain = subprocess.run([sys.executable, "bench.py", *common], cwd=ROOT, env=env, capture_output=True, text=True,
                           timeout=600)
    assert plain.returncode == 0, plain.stderr[-2000:]
    if launcher == "plain":
        cmd = [sys.executable, "bench.py", "--force-distributed", *common]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
               "127.0.0.1", "--master-port", str(_port()), "bench.py", "--gpus", "1", "--force-distributed", *common]
    one = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    a, b = _line(plain.stdout), _line(one.stdout)
    assert a["config"]["backend"] is None and a["config"]["collectives"] is None
    assert b["n_gpus"] == 1 and b["config"]["world"] == 1 and b["config"]["backend"] == "nccl"
    assert len(b["config"]["collectives"]) == 6
    assert b["config"]["winner_index_per_step"] == a["config"]["winner_index_per_step"]
    assert b["config"]["winner_f_last_step"] is not None and 0.0 <= b["config"]["winner_f_last_step"] < 1e-6
    assert b["value"] > 0

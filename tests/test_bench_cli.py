"""bench.py's rank accounting, without a GPU: `--gpus N` must either BE N ranks or refuse.  (The
runs themselves are -m gpu: tests/test_gpu_bench_ranks.py.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_gpus_flag_disagreeing_with_the_launcher_is_refused():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"], cwd=ROOT, capture_output=True, text=True,
                       env=_clean_env(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0
    assert "refusing" in r.stderr and "WORLD_SIZE=3" in r.stderr
    assert '{"metric"' not in r.stdout


def test_gpus_flag_without_a_launcher_spawns_that_many_ranks():
    """No GPU here: every spawned rank must stop at 'needs a GPU' -- two of them, from one command."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: the spawned two-rank run itself is tests/test_gpu_bench_ranks.py")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                       capture_output=True, text=True, env=_clean_env(), timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-3000:]
    assert '{"metric"' not in r.stdout


def test_inprocess_refuses_a_multi_rank_launcher():
    r = subprocess.run([sys.executable, "bench.py", "--inprocess", "--gpus", "2"], cwd=ROOT, capture_output=True,
                       text=True, env=_clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "one process" in r.stderr


def test_pmc_record_of_the_default_command_has_the_shape_bench_reads():
    """bench.py takes roofline.traffic and the f64 VALU scalars of the driver's command from profiles/: the record must sit
    under its command key as {"kernel_path": {...}} with the fields the line quotes (written flat once, the line said null)."""
    import json
    import bench
    with open(next(f for f in bench.PMC_FILES if os.path.exists(f))) as fh:  # (the newest round that has one)
        recs = json.load(fh)["commands"]
    key = "robot=panda,restarts=65536,steps=20,warmup=5,mode=speed,scaling=weak,targets=0,path=kernel,gpus=1"
    assert key in recs and "kernel_path" in recs[key]
    kp = recs[key]["kernel_path"]
    for f in ("hbm_bytes_per_restart", "f64_flops_per_restart", "launches", "valu_busy", "valu_active_lane_frac"):
        assert f in kp, f


def test_ranks_sharing_a_gpu_are_refused():
    """A line that says N GPUs needs N physical GPUs: distinct device indices; repeated UUIDs on distinct indices are
    reported, not fatal (a runtime that gives every device one placeholder must not cost the node its scaling run)."""
    import pytest
    import bench
    ok = [{"rank": r, "device": r, "uuid": f"GPU-{r:04x}"} for r in range(8)]
    assert bench.check_distinct_devices(ok, False) is None
    assert bench.check_distinct_devices([{"rank": 0, "device": 0, "uuid": None}, {"rank": 1, "device": 1, "uuid": None}], False) is None
    same_uuid = [dict(ok[0]), dict(ok[1], uuid=ok[0]["uuid"])]
    assert "repeated UUIDs" in bench.check_distinct_devices(same_uuid, False)
    shared = [dict(ok[0]), dict(ok[1], device=0)]
    with pytest.raises(SystemExit, match="share a physical GPU"):
        bench.check_distinct_devices(shared, False)
    assert bench.check_distinct_devices(shared, True) is None  # OPTIK_BENCH_ONE_DEVICE=1: the one-GPU test box

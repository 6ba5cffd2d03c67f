#!/usr/bin/env python3
"""Headline benchmark: random-restart IK solves/sec (Panda 7-DoF, tol_f = 1e-6).

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch: R = 65 536 restarts
(BASELINE.json config 2: Panda 7-DoF, SolutionMode::Speed) of one synthetic
reachable target, every restart run to termination (no early exit), followed by
the winner selection.  Targets, seeds x0 and all output buffers are resident in
HBM before the timed region.  With --path engine (default) the K timed steps are
submitted as K jobs and executed by one run of the streaming engine, which keeps
its slot pool full across step boundaries (continuous batching); --path kernel
launches one persistent solve kernel per step, back to back.  With N > 1 (one process per GPU under
torch.distributed.run) every rank solves its own contiguous restart range
[rank*R, (rank+1)*R) of the same target -- weak scaling, no data-path
collective -- and the per-step winner is chosen with one 8-byte RCCL
min-all-reduce of the selection key over xGMI.

The JSON line carries:
  roofline      algorithmic HBM bytes of the solve kernel / its mean duration
                (HIP events on the launch stream, recorded inside the C ABI)
  cpu_baseline  the CPU oracle (a port of the reference algorithm, NOT the
                reference binary) timed on this host's cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r1j_engine_pmc_traffic.json")


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same
    command (tools/profile_round.sh; FETCH_SIZE and WRITE_SIZE collected in separate runs).
    gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half the bytes of a wide
    coalesced read, so it is doubled; both counters are in KB."""
    try:
        with open(PMC_TRAFFIC_FILE) as fh:
            k = json.load(fh)["kernels"][kernel]
        return (2.0 * k["FETCH_SIZE"]["mean_kb_per_launch"] + k["WRITE_SIZE"]["mean_kb_per_launch"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None



ENGINE_POOL = 48  # steps whose restarts share one engine run


def load_chain(robot):
    """Flat chain table through the product's own URDF loader (C++, optik_robot_*)."""
    from optik_amd import Robot
    spec = {"panda": ("panda.urdf", "panda_link0", "panda_link8"),
            "ur10": ("ur10.urdf", "base_link", "ee_link")}[robot]
    path = os.path.join(ROOT, "optik_amd", "robots", spec[0])
    return Robot.from_urdf_file(path, spec[1], spec[2])


def usable_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box exposes 256 logical CPUs under a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(robot_name, chain_tables, target7, x0, seconds_budget=15.0):
    """Times the CPU oracle (checker code, used here only as the reported baseline)."""
    from oracle import binding as ob
    ob.build()
    ch = ob.make_chain(**chain_tables)
    cfg = ob.make_config(solution_mode="speed", tol_f=1e-6)
    cores = usable_cores()
    # calibrate on a small sample, then size the timed sample to ~seconds_budget
    t0 = time.perf_counter()
    ob.ik(ch, cfg, target7, x0, 1, 1 + 64 * cores, n_threads=cores, early_exit=False)
    rate = 64 * cores / max(time.perf_counter() - t0, 1e-6)
    n = int(min(max(rate * seconds_budget, 256 * cores), 4_000_000))
    t0 = time.perf_counter()
    res = ob.ik(ch, cfg, target7, x0, 0, n, n_threads=cores, early_exit=False)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "restarts/s", "cores": cores, "kind": "port",
            "sample": f"{robot_name}: restarts 0..{n - 1} of the bench target, all run to termination, "
                      f"{cores} threads pulling indices from a shared counter, {dt:.1f} s",
            "winner": int(res["winner"]) if res["found"] else -1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48,
                    help="timed steps; on the engine path they are pooled into one run (one drain of the slot pool)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--restarts", type=int, default=65536, help="restarts per GPU per step")
    ap.add_argument("--robot", default="panda", choices=["panda", "ur10"])
    ap.add_argument("--path", default="auto", choices=["auto", "engine", "kernel"],
                    help="engine: streaming phase kernels with continuous batching (steps submitted "
                         "together share the slot pool); kernel: one persistent solve kernel per step; "
                         "auto (default) = engine: one 65 536-restart step takes 12 ms in an engine run "
                         "(the tail kernel finishes the longest restarts) against 14 ms for a solve-kernel "
                         "launch, and 2.4 ms per step from there on; results are identical")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    if args.path == "auto":
        args.path = "engine"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: optik_amd has no CPU fallback")
    # OPTIK_BENCH_BACKEND=gloo + OPTIK_BENCH_ONE_DEVICE=1 exercise the multi-rank path on a
    # single-GPU box (all ranks share cuda:0); the real runs use nccl (= RCCL) and one GPU per rank
    backend = os.environ.get("OPTIK_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("OPTIK_BENCH_ONE_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)

    from optik_amd import _native as nat
    from optik_amd.parallel import shard_range, select_winner

    robot = load_chain(args.robot)
    hc = robot.hip_chain(dev)
    n = robot.num_positions()
    R = args.restarts
    K, W = args.steps, args.warmup

    # synthetic workload: reachable targets FK(q*), q* and x0 uniform in the limits
    rng = np.random.default_rng(0)
    lb, ub = (np.array(v) for v in robot.joint_limits())
    n_tgt = K + W
    q_star = rng.uniform(lb, ub, size=(n_tgt, n))
    x0_host = rng.uniform(lb, ub, size=(n_tgt, n))
    pose = hc.fk_batch(torch.tensor(q_star.T.copy(), device=dev))  # [7, n_tgt] on the GPU
    targets = pose.T.contiguous()
    x0 = torch.tensor(x0_host, device=dev)
    cfg = nat.make_config(solution_mode="speed", tol_f=1e-6)
    begin, end = shard_range(0, R * world, rank, world)
    n_buf = max(K, W) if args.path == "engine" else 1
    bufs = [hc.alloc_ik_buffers(1, R, per_restart=True) for _ in range(n_buf)]
    # the per-step winner records are rows of two tensors, so that the winners of a whole run are
    # selected (and, with several ranks, reduced) in one piece without gathering them first
    win_idx_all = torch.zeros(n_buf, dtype=torch.int64, device=dev)
    win_key_all = torch.zeros(n_buf, dtype=torch.float64, device=dev)
    for k, b in enumerate(bufs):
        b["win_idx"] = win_idx_all[k:k + 1]
        b["win_key"] = win_key_all[k:k + 1]
    if args.path == "engine":
        hc.engine_reserve()  # the slot pool: allocated with the other buffers, not inside a run
    # plumbing first-use costs (torch's lazily loaded elementwise kernels, the communicator of the
    # first collective) are paid here on dummy records, not inside the timed region when W = 0
    select_winner({"win_idx": torch.zeros(1, dtype=torch.int64, device=dev),
                   "win_key": torch.zeros(1, dtype=torch.float64, device=dev)}, "speed", distributed)
    torch.cuda.synchronize()

    def run_steps(first, count):
        """`count` steps starting at target `first`; returns the per-step global winners."""
        if args.path == "engine":
            # every step is its own job (own target, own outputs); jobs submitted together
            # share the engine's slot pool, then one blocking run executes them all
            # (at most ENGINE_POOL jobs per run: the engine's job table holds 64)
            for g0 in range(0, count, ENGINE_POOL):
                for k in range(g0, min(g0 + ENGINE_POOL, count)):
                    i = first + k
                    hc.engine_submit(cfg, targets[i:i + 1], x0[i:i + 1], begin, end, bufs=bufs[k])
                hc.engine_run()
            stacked = {"win_idx": win_idx_all[:count], "win_key": win_key_all[:count]}
            return select_winner(stacked, "speed", distributed)  # one collective for all steps
        winners = []
        for k in range(count):
            i = first + k
            hc.ik_batch(cfg, targets[i:i + 1], x0[i:i + 1], begin, end, bufs=bufs[0])
            winners.append(select_winner(bufs[0], "speed", distributed))
        return torch.cat(winners)

    if W:
        run_steps(0, W)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    hc.set_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    winners = run_steps(W, K)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        from optik_amd.parallel import _all_reduce
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        _all_reduce(tmax, dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    last = bufs[K - 1] if args.path == "engine" else bufs[0]
    n_success = int((last["status"] == nat.RES_STOPVAL).sum().item())
    mean_evals = float(last["evals"].double().mean().item())

    if rank == 0:
        total = float(R) * world * K
        out_bytes = 8 * n + 8 + 8 + 4 + 4  # x[n] + f + key + status + evals written per restart
        if args.path == "engine":
            st = hc.engine_stats()
            trips = int(nat.lib().optik_hip_engine_last_trips(hc._h))
            per_kernel = {k: st[k + "_ms"] for k in ("eval", "update", "nnls", "finish")}
            dom = max(per_kernel, key=per_kernel.get)
            m = n + 1
            # algorithmic HBM bytes of one launch of the dominant kernel
            if dom == "nnls":
                units = st["nnls_problems"] / max(st["launches"], 1)     # sub-problems per launch
                # packed record in (rows of E^-1 + h), multipliers + {mode, rnorm} out
                unit_bytes = 8 * (n * (n + 1) // 2 + 2 * n) + 8 * (2 * n + 2)
            else:
                slots_per_launch = float(R) * K / max(st["launches"], 1) * mean_evals  # slot-trips per launch (approx.)
                units = slots_per_launch
                nl = n * (n + 1) // 2
                # planes read + written per slot-trip (update also writes the packed problem record,
                # finish reads it back with the multipliers)
                unit_bytes = {"eval": 8 * (2 * n + 6), "update": 8 * (3 * nl + 11 * n + 8),
                              "finish": 8 * (2 * nl + 10 * n + 12)}[dom]
            kernel_ms = per_kernel[dom]
            achieved = unit_bytes * units / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
            kname = {"eval": "eng_eval_kernel", "update": "eng_update_kernel", "nnls": "eng_nnls_coop_kernel",
                     "finish": "eng_finish_kernel"}[dom]
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(kname),
                    "traffic_source": os.path.relpath(PMC_TRAFFIC_FILE, ROOT), "kernel": kname,
                    "kernel_ms": kernel_ms, "launches_timed": st["sampled_trips"],
                    "algorithmic_bytes_per_launch": unit_bytes * units,
                    "algorithmic_bytes_per_unit": unit_bytes, "units_per_launch": units,
                    "unit_name": "bounded sub-problem" if dom == "nnls" else "slot-trip",
                    "all_kernels_ms": per_kernel, "trips": trips, "sub_pools": st["pools"],
                    "launches": st["launches"], "restart_output_bytes": out_bytes,
                    # the whole path at its boundary: SURVEY 8d's per-restart figure with in-kernel
                    # seeds (outputs only) and with the seeds counted as read (16n + 16)
                    "path_boundary": {"bytes_per_restart": out_bytes, "bytes_per_restart_survey": 16 * n + 16,
                                      "GBps": total / elapsed * out_bytes / 1e9,
                                      "frac": total / elapsed * out_bytes / 1e9 / HBM_PEAK_GBS}}
            info = {"grid": None, "block": 128, "lds_bytes": 0}
        else:
            kernel_ms, launches = hc.timing_mean()
            info = hc.last_launch()
            achieved = out_bytes * R / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "ik_solve_kernel",
                    "kernel_ms": kernel_ms, "launches_timed": launches,
                    "algorithmic_bytes_per_unit": out_bytes, "units_per_launch": R}
        line = {
            "metric": "random-restart IK solves/sec (Panda 7-DoF, 1e-6 tol)" if args.robot == "panda"
                      else f"random-restart IK solves/sec ({args.robot}, 1e-6 tol)",
            "value": total / elapsed,
            "unit": "restarts/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.robot} {n}-DoF, {R} random restarts per GPU per step, one target "
                                   "per step, SolutionMode::Speed, every restart run to termination",
                       "path": args.path, "restarts_per_gpu": R, "tol_f": 1e-6,
                       "parallelism": f"restart-range x{world}",
                       "success_rate_last_step": n_success / R, "mean_evals_per_restart": mean_evals,
                       "objective_gradient_evals_per_s": total / elapsed * mean_evals,
                       # global winner (lowest successful restart index) of every timed step: the
                       # same for any number of ranks covering the same restart range
                       "winner_index_per_step": [int(v) for v in winners.cpu().tolist()][:64],
                       "grid": info["grid"], "block": info["block"], "lds_bytes": info["lds_bytes"]},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            tables = robot.chain_tables()
            line["cpu_baseline"] = cpu_baseline(args.robot, tables, targets[W].cpu().numpy(),
                                                x0_host[W], args.cpu_seconds)
            line["cpu_baseline"]["gpu_winner_same_target"] = int(winners[0].item())
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

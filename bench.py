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
launches one persistent solve kernel per step, back to back.

Multi-GPU.  `python bench.py --gpus N` with no launcher around it starts the N ranks itself: it
re-executes under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1` with the same arguments (one process per GPU, RCCL over xGMI); under an outer launcher
(RANK / WORLD_SIZE in the environment: what the driver uses for N > 1) it is one of the ranks.  The
line's "n_gpus" is the number of ranks that actually joined the process group, and config carries
the world size, the backend string and every rank's torch.cuda.current_device(); a --gpus that
disagrees with WORLD_SIZE is refused.  `--inprocess` drives the other multi-GPU form instead: ONE
process, N devices behind the C ABI (optik_robot_set_devices: a host thread per device, host min over
N 16-byte records), through Robot.ik with SolutionMode::Quality and a restart budget of R x N.
Workload flags (one process per GPU):
  --scaling weak    (default) every rank solves its own contiguous restart range
                    [rank*R, (rank+1)*R) of the step's target: R restarts per GPU per step
  --scaling strong  the step's R restarts are cut into one contiguous range per rank
                    (BASELINE.json config 4: 4 M restarts sharded 8-way)
  --mode speed|quality   SolutionMode: the per-step winner is the lowest successful index
                    (one 8-byte min-all-reduce) or the success closest to the seed (min-all-reduce
                    of ||x - x0||, then of the index among the ranks holding the minimum)
  --targets T       BASELINE.json config 5: a step is T independent targets x --restarts
                    restart indices each (Speed with early exit, as Robot::ik), the targets cut
                    into one contiguous part per rank, no collective; value = ik() calls/s.
                    Runs on the cooperative kernel with restart-major hand-out by default (what
                    Robot.ik_batch picks for a Speed batch); --path engine for the engine

The JSON line carries:
  roofline      algorithmic HBM bytes of the dominant kernel / its mean duration (HIP events
                attached to that kernel's dispatches inside the C ABI); `traffic` = PMC bytes per
                launch when profiles/ holds a PMC pass of this exact command, else null;
                `secondary` = the f64 vector-ALU view (HBM is not what binds this path)
  cpu_baseline  the CPU oracle (a port of the reference algorithm, NOT the reference binary)
                timed on this host's cores (1 thread, half, all) on a bounded sample
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_VALU_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 f64 lanes/clk x 2 (FMA) x 2.4 GHz
# ... and what this path can reach at best: -ffp-contract=off (rustc never fuses a*b+c, and the
# bit-exactness contract with the oracle forbids it) makes every f64 instruction ONE flop
F64_VALU_NOFMA_TFLOPS = 39.3
PMC_FILES = [os.path.join(ROOT, "profiles", "r4_pmc_by_command.json"),
             os.path.join(ROOT, "profiles", "r3_pmc_by_command.json"),
             os.path.join(ROOT, "profiles", "r2_pmc_by_command.json")]
PMC_FILE = PMC_FILES[0]

ENGINE_POOL = 240  # steps whose restarts share one engine run (the engine pools up to 256 jobs)


def command_key(args, world):
    """Identifies the workload a PMC pass was taken on (tools/profile_round.sh writes the same key)."""
    return (f"robot={args.robot},restarts={args.restarts},steps={args.steps},warmup={args.warmup},"
            f"mode={args.mode},scaling={args.scaling},targets={args.targets},path={args.path},gpus={world}")


def pmc_for(key):
    """The PMC record of this exact command (newest profile file first), with the file it came from."""
    global PMC_FILE
    for f in PMC_FILES:
        try:
            with open(f) as fh:
                rec = json.load(fh)["commands"].get(key)
        except (OSError, KeyError, ValueError):
            rec = None
        if rec:
            PMC_FILE = f
            return rec
    return None


def load_chain(robot):
    """Flat chain table through the product's own URDF loader (C++, optik_robot_*)."""
    from optik_amd import Robot
    spec = {"panda": ("panda.urdf", "panda_link0", "panda_link8"),
            "ur10": ("ur10.urdf", "base_link", "ee_link"),
            # a synthetic 8-DoF chain (tests/golden/robots/): the largest n the kernels are built for
            "arm8": (os.path.join("..", "..", "tests", "golden", "robots", "arm8.urdf"), "l0", "l9"),
            # synthetic 10- and 16-joint chains: the general kernels of ik_wide.hpp (use --path kernel)
            "arm10": (os.path.join("..", "..", "tests", "golden", "robots", "arm10.urdf"), "l0", "l11"),
            "arm16": (os.path.join("..", "..", "tests", "golden", "robots", "arm16.urdf"), "l0", "l17")}[robot]
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


def cpu_baseline(robot_name, chain_tables, target7, x0, mode, seconds_budget=15.0):
    """Times the CPU oracle (checker code, used here only as the reported baseline) at 1 thread,
    half the usable cores (the reference's advice, README.md:90-91) and all of them."""
    from oracle import binding as ob
    flags = ob.use_native_build()  # -O3 -march=native of THIS host, same arithmetic
    ch = ob.make_chain(**chain_tables)
    cfg = ob.make_config(solution_mode=mode, tol_f=1e-6)
    cores = usable_cores()
    points = sorted({1, max(1, cores // 2), cores})
    share = {1: 0.25, max(1, cores // 2): 0.3, cores: 0.45}
    by_threads, winner, sample = {}, -1, []
    for th in points:
        budget = seconds_budget * share.get(th, 0.3) if len(points) > 1 else seconds_budget
        t0 = time.perf_counter()
        ob.ik(ch, cfg, target7, x0, 1, 1 + 32 * th, n_threads=th, early_exit=False)
        rate = 32 * th / max(time.perf_counter() - t0, 1e-6)
        n = int(min(max(rate * budget, 64 * th), 4_000_000))
        t0 = time.perf_counter()
        res = ob.ik(ch, cfg, target7, x0, 0, n, n_threads=th, early_exit=False)
        dt = time.perf_counter() - t0
        by_threads[str(th)] = n / dt
        sample.append(f"{th} thread(s): restarts 0..{n - 1} in {dt:.1f} s")
        if th == cores:
            winner = int(res["winner"]) if res["found"] else -1
    return {"value": by_threads[str(cores)], "unit": "restarts/s", "cores": cores, "kind": "port",
            "by_threads": by_threads, "build": flags,
            "sample": f"{robot_name}: the bench target, SolutionMode {mode}, every restart run to termination, threads "
                      f"pulling indices from a shared counter; " + "; ".join(sample),
            "winner": winner}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run the N ranks under torch.distributed.run
    (one process per GPU) with the same arguments and return its exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def inprocess_main(args):
    """One process, G devices behind the C ABI (SURVEY 8e's other form; robot_host.cpp:
    optik_robot_set_devices): a step is Robot.ik(SolutionMode::Quality, max_restarts = R x G) on one
    target -- every restart of [0, R x G) runs to termination, device g takes the g-th contiguous
    part from its own host thread, the host keeps the minimum of the G (key, index) records."""
    from optik_amd import SolverConfig
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: optik_amd has no CPU fallback")
    G = args.gpus or 1
    K, W = args.steps, args.warmup
    R = args.restarts or 65536
    have = torch.cuda.device_count()
    one = os.environ.get("OPTIK_BENCH_ONE_DEVICE") == "1"
    devices = [0] * G if one else list(range(G))
    if not one and G > have:
        raise SystemExit(f"--inprocess --gpus {G}: this node has {have} device(s)")
    robot = load_chain(args.robot)
    robot.set_devices(devices)
    n = robot.num_positions()
    rng = np.random.default_rng(0)
    lb, ub = (np.array(v) for v in robot.joint_limits())
    q_star = rng.uniform(lb, ub, size=(K + W, n))
    x0 = rng.uniform(lb, ub, size=(K + W, n))
    targets = [robot.fk(q) for q in q_star]
    cfg = SolverConfig(solution_mode="quality", max_time=0.0, max_restarts=R * G, tol_f=1e-6)

    parts_used = []  # devices every timed call was actually cut over (optik_robot_last_parts)

    def run_steps(first, count):
        out = []
        for k in range(first, first + count):
            r = robot.ik(cfg, targets[k], x0[k], return_index=True)
            out.append(-1 if r is None else int(r[2]))
            parts_used.append(robot.last_parts())
        return out

    if W:
        run_steps(0, W)
    for d in set(devices):
        torch.cuda.synchronize(d)
    rep_elapsed, winners = [], None
    for _rep in range(max(1, args.reps)):
        t0 = time.perf_counter()
        w_rep = run_steps(W, K)  # Robot.ik blocks until the G parts are done and reduced
        rep_elapsed.append(time.perf_counter() - t0)
        if winners is not None and winners != w_rep:
            raise SystemExit("winners differ between repetitions of the same steps")
        winners = w_rep
    elapsed = sorted(rep_elapsed)[len(rep_elapsed) // 2]
    total = float(R) * G * K
    out_bytes = 8 * n + 8 + 8 + 4 + 4
    line = {
        "metric": "random-restart IK solves/sec (Panda 7-DoF, 1e-6 tol)" if args.robot == "panda"
                  else f"random-restart IK solves/sec ({args.robot}, 1e-6 tol)",
        "value": total / elapsed, "unit": "restarts/s", "n_gpus": min(parts_used[-K:]) if parts_used else 0, "steps": K, "warmup": W,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.robot} {n}-DoF, {R} random restarts per GPU per step (one Robot.ik call over "
                               f"[0, {R * G})), one target per step, SolutionMode::Quality, every restart run to termination",
                   "inprocess": True, "devices": devices, "devices_configured": robot.num_devices(),
                   # (n_gpus above is what the calls were really cut over -- the host API keeps a range that is not worth
                   # cutting on one device -- not what was configured)
                   "parts_per_call": sorted(set(parts_used[-K:])), "world": 1,
                   "backend": "host threads + host min (no collective)",
                   "reps": len(rep_elapsed), "rep_reported": "median", "value_reps": [total / e for e in rep_elapsed],
                   "restarts_per_gpu": R, "tol_f": 1e-6, "solution_mode": "quality",
                   "parallelism": f"restart-range x{G} (in-process)", "winner_index_per_step": winners[:64]},
        # this form goes through the host API, which does not expose per-kernel timers: the boundary view only
        "roofline": {"bound": "hbm", "achieved": total / elapsed * out_bytes / 1e9, "peak": HBM_PEAK_GBS * G,
                     "unit": "GB/s", "frac": total / elapsed * out_bytes / 1e9 / (HBM_PEAK_GBS * G), "traffic": None,
                     "kernel": None, "algorithmic_bytes_per_unit": out_bytes, "unit_name": "restart (path boundary)"},
        "cpu_baseline": None,
    }
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs (= ranks) of this node; without a launcher N > 1 ranks are spawned here")
    ap.add_argument("--inprocess", action="store_true",
                    help="one process driving --gpus devices through optik_robot_set_devices (Robot.ik, "
                         "Quality, restart budget R x N) instead of one process per GPU")
    ap.add_argument("--reps", type=int, default=5,
                    help="repetitions of the K-step timed run: value / ms_per_step are the MEDIAN repetition's, "
                         "config.value_reps lists all of them")
    ap.add_argument("--steps", type=int, default=48,
                    help="timed steps; on the engine path they are pooled into one run (one drain of the slot pool)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--restarts", type=int, default=None,
                    help="restarts per step: per GPU (weak) or in total (strong); per target with --targets; "
                         "default 65536 (256 with --targets)")
    ap.add_argument("--robot", default="panda", choices=["panda", "ur10", "arm8", "arm10", "arm16"])
    ap.add_argument("--mode", default="speed", choices=["speed", "quality"], help="SolutionMode (config.rs:3-8)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--targets", type=int, default=0,
                    help="config 5: targets per step (cut into one part per rank), each an independent ik() call")
    ap.add_argument("--path", default="auto", choices=["auto", "engine", "kernel"],
                    help="kernel (= auto, the default): ONE persistent solve kernel for the run's steps, restart state "
                         "in registers and LDS; engine: streaming phase kernels over an HBM slot pool with "
                         "continuous batching; results are identical")
    ap.add_argument("--find-any", action="store_true",
                    help="--targets: the reference's default reading of should_exit (rayon find_any, lib.rs:409-412: ANY "
                         "success ends a target's other restarts) as Robot.ik_batch runs it unless set_parallelism(1); the "
                         "default here is the deterministic reading (lowest solved index), which every rank count reproduces")
    ap.add_argument("--force-distributed", action="store_true",
                    help="run the process-group path (init_process_group, the min / sum / max all-reduces, "
                         "all_gather_object) even with ONE rank: RCCL on a single GPU")
    ap.add_argument("--set-option", action="append", default=[], metavar="NAME=VALUE",
                    help="a tuning option of the kernel layer for this run (include/optik_hip.h: optik_hip_set_option), "
                         "e.g. engine_pools=2; experiments only -- the line records what was set")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    if args.path == "auto":
        # the single-launch solvers (state in registers and LDS, one launch per run): from one full load of the
        # chip on the lane-per-restart form (ik_lane64.hpp), below it the quad solver; `--path engine` is the
        # streaming engine of rounds 1-3 (state in an HBM slot pool, five phase kernels per trip)
        args.path = "kernel"
    if args.restarts is None:
        args.restarts = 256 if args.targets else 65536

    env_world = os.environ.get("WORLD_SIZE")
    if args.inprocess:
        if env_world is not None and int(env_world) > 1:
            raise SystemExit("--inprocess is one process driving every device: do not launch it under a multi-rank launcher")
        return inprocess_main(args)
    if env_world is None and (args.gpus or 1) > 1:
        # no launcher around us: start the ranks ourselves
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if args.gpus is not None and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to print a "
                         f"line whose n_gpus is not the number of ranks that ran")
    distributed = world > 1 or args.force_distributed
    if distributed and env_world is None:
        # one rank without a launcher: the rendezvous a launcher would have set up
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
        s_.close()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
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
        if dist.get_world_size() != world:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, WORLD_SIZE says {world}")
        # proof that N ranks joined, each on its own GPU: (rank, current device, device name, pid)
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, {"rank": rank, "device": int(torch.cuda.current_device()),
                                              "name": torch.cuda.get_device_name(dev_index), "pid": os.getpid()})
        backend_name = str(dist.get_backend())
    else:
        rank_devices = [{"rank": 0, "device": int(torch.cuda.current_device()),
                         "name": torch.cuda.get_device_name(dev_index), "pid": os.getpid()}]
        backend_name = None

    from optik_amd import _native as nat
    from optik_amd.parallel import I64_MAX, shard_range, select_winner, gather_winner_x
    for kv in args.set_option:
        name, _, val = kv.partition("=")
        nat.set_option(name, int(val) if val.lstrip("-").isdigit() else val)

    robot = load_chain(args.robot)
    hc = robot.hip_chain(dev)
    n = robot.num_positions()
    R = args.restarts
    K, W = args.steps, args.warmup
    T = args.targets
    mode = args.mode

    # synthetic workload: reachable targets FK(q*), q* and x0 uniform in the limits
    rng = np.random.default_rng(0)
    lb, ub = (np.array(v) for v in robot.joint_limits())
    per_step = max(T, 1)
    n_tgt = (K + W) * per_step
    q_star = rng.uniform(lb, ub, size=(n_tgt, n))
    x0_host = rng.uniform(lb, ub, size=(n_tgt, n))
    pose = hc.fk_batch(torch.tensor(q_star.T.copy(), device=dev))  # [7, n_tgt] on the GPU
    targets = pose.T.contiguous()
    x0 = torch.tensor(x0_host, device=dev)
    cfg = nat.make_config(solution_mode=mode, tol_f=1e-6)
    if T:
        t_lo, t_hi = shard_range(0, T, rank, world)    # this rank's targets of every step
        begin, end = 0, R                              # every target runs restart indices 0..R-1
        T_loc = t_hi - t_lo
        cols = R
    else:
        t_lo, t_hi, T_loc = 0, 1, 1
        begin, end = shard_range(0, R * world if args.scaling == "weak" else R, rank, world)
        cols = end - begin
    if T_loc < 1 or cols < 1:
        raise SystemExit("more ranks than work items: nothing to do on this rank")
    n_buf = max(K, W) if args.path == "engine" else 1
    # kernel path: the steps of a run are the targets of ONE launch of the solve kernel (each step its own
    # target, seed and output columns) -- the persistent waves pull (step, restart) items from one queue,
    # so the run pays one fill and one drain of the chip instead of one per step (the engine path pools
    # its steps the same way: one run of the slot pool)
    pooled_kernel = args.path == "kernel" and not T
    bufs = [hc.alloc_ik_buffers(T_loc, cols, per_restart=not T) for _ in range(n_buf)]
    # (one set of output buffers per run length: the timed run of K steps and the warm-up run of W)
    kbufs = {c: hc.alloc_ik_buffers(c, cols) for c in {K, W} if c} if pooled_kernel else {}
    # the per-step winner records are rows of two tensors, so that the winners of a whole run are
    # selected (and, with several ranks, reduced) in one piece without gathering them first
    win_idx_all = torch.zeros((n_buf, T_loc), dtype=torch.int64, device=dev)
    win_key_all = torch.zeros((n_buf, T_loc), dtype=torch.float64, device=dev)
    win_x_all = torch.zeros((n_buf, T_loc, n), dtype=torch.float64, device=dev)
    win_f_all = torch.zeros((n_buf, T_loc), dtype=torch.float64, device=dev)
    for k, b in enumerate(bufs):
        b["win_idx"] = win_idx_all[k]
        b["win_key"] = win_key_all[k]
        b["win_x"] = win_x_all[k]
        b["win_f"] = win_f_all[k]
    win_xf = {}  # the global winners' x and f of the last run (every rank holds them, as Robot::ik returns them)

    def exchange(rec, count):
        """The cross-rank half of a step (lib.rs:397-413 over the ranks): two 8-byte min-all-reduces pick the
        winner, a sum-all-reduce of <= 64 B per target hands its x and f to every rank."""
        win = select_winner(rec, mode, distributed)
        if distributed:
            win_xf["x"], win_xf["f"] = gather_winner_x(rec, win, begin, end, True)
        return win
    if args.path == "engine":
        hc.engine_reserve()  # the slot pool: allocated with the other buffers, not inside a run
    # plumbing first-use costs (torch's lazily loaded elementwise kernels, the communicator of the
    # first collective) are paid here on dummy records, not inside the timed region when W = 0
    select_winner({"win_idx": torch.zeros(1, dtype=torch.int64, device=dev),
                   "win_key": torch.zeros(1, dtype=torch.float64, device=dev)}, mode, distributed and not T)
    torch.cuda.synchronize()
    flags = nat.IK_EARLY_EXIT if T else 0
    if T and args.path == "kernel":
        flags |= nat.IK_RESTART_MAJOR  # every target's low restart indices first

    def run_steps(first, count):
        """`count` steps starting at step `first`; returns the per-step winners ([count, T_loc])."""
        if args.path == "engine":
            # every step is its own job (own targets, own outputs); jobs submitted together
            # share the engine's slot pool, then one blocking run executes them all
            for g0 in range(0, count, ENGINE_POOL):
                for k in range(g0, min(g0 + ENGINE_POOL, count)):
                    i = (first + k) * per_step + t_lo
                    hc.engine_submit(cfg, targets[i:i + T_loc], x0[i:i + T_loc], begin, end, flags=flags,
                                     bufs=bufs[k])
                hc.engine_run()
            stacked = {"win_idx": win_idx_all[:count].reshape(-1), "win_key": win_key_all[:count].reshape(-1),
                       "win_x": win_x_all[:count].reshape(-1, n), "win_f": win_f_all[:count].reshape(-1)}
            # config 5: every rank owns its targets outright -- no collective
            if T:
                return select_winner(stacked, mode, False).reshape(count, T_loc)
            return exchange(stacked, count).reshape(count, T_loc)
        if pooled_kernel:
            i = first + t_lo
            kb = kbufs[count]
            hc.ik_batch(cfg, targets[i:i + count], x0[i:i + count], begin, end, flags=flags, bufs=kb, per_restart=True)
            return exchange(kb, count).clone().reshape(count, 1)
        winners = []
        for k in range(count):
            i = (first + k) * per_step + t_lo
            if T and mode == "speed":
                winners.append(batch_rounds(targets[i:i + T_loc], x0[i:i + T_loc]))
                continue
            hc.ik_batch(cfg, targets[i:i + T_loc], x0[i:i + T_loc], begin, end, flags=flags, bufs=bufs[0],
                        per_restart=not T)
            winners.append((select_winner(bufs[0], mode, False) if T else exchange(bufs[0], 1)).clone())
        return torch.stack(winners)

    def batch_rounds(tg, xs):
        """One Speed batch of independent ik() calls as the product schedules it (robot_host.cpp:ik_batch_on_device):
        a latency-sized first round of 128 restart indices per target with early exit and restart-major hand-out,
        then rounds four times as long for the targets still unsolved (they drop out as they are solved)."""
        win = torch.full((tg.shape[0],), I64_MAX, dtype=torch.int64, device=dev)
        live = torch.arange(tg.shape[0], device=dev)
        b, rnd = begin, 128
        while b < end and live.numel():
            e = min(end, b + rnd)
            out = hc.ik_batch(cfg, tg[live].contiguous(), xs[live].contiguous(), b, e,
                              flags=nat.IK_EARLY_EXIT | (nat.IK_RESTART_MAJOR if b < 256 else 0)
                              | (nat.IK_FIND_ANY if args.find_any else 0), per_restart=False)
            solved = out["win_idx"] >= 0
            win[live[solved]] = out["win_idx"][solved]
            live = live[~solved]
            b, rnd = e, rnd * 4
        return win

    if W:
        run_steps(0, W)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    # The K timed steps are repeated `--reps` times (same steps, same buffers); every repetition is
    # bracketed by a barrier + synchronize on both sides and reduced to the MAX over ranks.  The line
    # reports the median repetition (ms_per_step x steps = that one repetition), and all of them in
    # config.value_reps: box-to-box and run-to-run spread is a few percent, more than some rounds move.
    rep_elapsed = []
    winners = None
    hc.set_timing(True)  # (HIP events around the dominant kernel's launches of ALL the timed repetitions)
    for _rep in range(max(1, args.reps)):
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w_rep = run_steps(W, K)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if distributed:
            from optik_amd.parallel import _all_reduce
            tmax = torch.tensor([el], dtype=torch.float64, device=dev)
            _all_reduce(tmax, dist.ReduceOp.MAX)
            el = float(tmax.item())
        rep_elapsed.append(el)
        if winners is not None and not torch.equal(winners, w_rep) and not (T and args.find_any):
            raise SystemExit("winners differ between repetitions of the same steps")
        winners = w_rep
    elapsed = sorted(rep_elapsed)[len(rep_elapsed) // 2]  # the median repetition (upper median for even counts)

    last = bufs[K - 1] if args.path == "engine" else bufs[0]
    if pooled_kernel:  # the last timed step's columns of the pooled launch
        last = dict(status=kbufs[K]["status"][(K - 1) * cols:K * cols], evals=kbufs[K]["evals"])
    solved_targets = int((winners >= 0).logical_and(winners < torch.iinfo(torch.int64).max).sum().item())
    if T:
        n_success, mean_evals, mean_exec = None, None, None
    else:
        n_success = int((last["status"] == nat.RES_STOPVAL).sum().item())
        # NLopt's count over every restart of the timed steps (the population the executed count covers)
        timed = bufs[:K] if args.path == "engine" else bufs
        if pooled_kernel:
            mean_evals = float(last["evals"].double().mean().item())
        else:
            mean_evals = float(sum(b["evals"].double().sum().item() for b in timed) / (len(timed) * cols))
        mean_exec = None

    if rank == 0:
        units_per_step = float(T) if T else float(cols) * world      # ik() calls or restarts, all ranks
        total = units_per_step * K
        out_bytes = 8 * n + 8 + 8 + 4 + 4  # x[n] + f + key + status + evals written per restart
        key = command_key(args, world)
        pmc = pmc_for(key)
        if args.path == "engine":
            st = hc.engine_stats()
            trips = int(nat.lib().optik_hip_engine_last_trips(hc._h))
            per_kernel = {k: st[k + "_ms"] for k in ("eval", "update", "nnls", "finish")}
            dom = max(per_kernel, key=per_kernel.get)
            if st.get("evals_executed") and not T:
                mean_exec = st["evals_executed"] / (float(cols) * K)
            # algorithmic HBM bytes of one launch of the dominant kernel
            if dom == "nnls":
                units = st["nnls_problems"] / max(st["launches"], 1)     # sub-problems per launch
                # packed record in (rows of E^-1 + h), multipliers + {mode, rnorm} out
                unit_bytes = 8 * (n * (n + 1) // 2 + 2 * n) + 8 * (2 * n + 2)
            else:
                slot_trips = st.get("slot_trips") or float(cols) * K * (mean_evals or 39.0)
                units = slot_trips / max(st["launches"], 1)              # slot-trips per launch
                nl = n * (n + 1) // 2
                # planes read + written per slot-trip (update also writes the packed problem record,
                # finish reads it back with the multipliers)
                unit_bytes = {"eval": 8 * (2 * n + 6), "update": 8 * (3 * nl + 11 * n + 8),
                              "finish": 8 * (2 * nl + 10 * n + 12)}[dom]
            kernel_ms = per_kernel[dom]
            achieved = unit_bytes * units / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
            kname = {"eval": "eng_eval_kernel", "update": "eng_update_kernel", "nnls": "eng_nnls_coop_kernel",
                     "finish": "eng_finish_kernel"}[dom]
            traffic, traffic_note = None, f"no PMC pass of this command in {os.path.relpath(PMC_FILE, ROOT)} (key: {key})"
            if pmc and kname in pmc.get("kernels", {}):
                kp = pmc["kernels"][kname]
                # gfx950 (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half the bytes of a wide
                # coalesced read, so it is doubled; both counters are in KB; per launch of the timed run
                traffic = (2.0 * kp["FETCH_SIZE_kb_per_launch"] + kp["WRITE_SIZE_kb_per_launch"]) * 1024.0
                traffic_note = (f"{os.path.relpath(PMC_FILE, ROOT)}: separate FETCH_SIZE / WRITE_SIZE passes of this "
                                f"command, the {kp['launches']} launches of the timed run")
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                    "kernel": kname, "kernel_ms": kernel_ms, "launches_timed": st["sampled_trips"],
                    "algorithmic_bytes_per_launch": unit_bytes * units,
                    "algorithmic_bytes_per_unit": unit_bytes, "units_per_launch": units,
                    "unit_name": "bounded sub-problem" if dom == "nnls" else "slot-trip",
                    "all_kernels_ms": per_kernel, "trips": trips, "sub_pools": st["pools"],
                    # who finishes the run's last restarts once the queue is dry (ik_quad_tail.hpp)
                    "tail": dict(zip(("solver", "restarts_taken_over"),
                                     (lambda sv, nr: ({0: "none", 3: "quad solver"}[sv], nr))(*hc.engine_last_tail()))),
                    "launches": st["launches"], "restart_output_bytes": out_bytes,
                    # the whole path at its boundary: SURVEY 8d's per-restart figure with in-kernel
                    # seeds (outputs only) and with the seeds counted as read (16n + 16)
                    "path_boundary": None if T else {
                        "bytes_per_restart": out_bytes, "bytes_per_restart_survey": 16 * n + 16,
                        "GBps": total / elapsed * out_bytes / 1e9,
                        "frac": total / elapsed * out_bytes / 1e9 / HBM_PEAK_GBS}}
            # HBM is the roofline the north star names, but this path is bound by f64 vector
            # arithmetic and its latency: the secondary view prices it against the f64 VALU peak
            if pmc and pmc.get("f64_flops_per_restart") and not T:
                fl = pmc["f64_flops_per_restart"]
                tf = total / elapsed * fl / 1e12
                roof["secondary"] = {"bound": "valu_f64", "achieved": tf, "peak": F64_VALU_PEAK_TFLOPS,
                                     "unit": "TFLOP/s", "frac": tf / F64_VALU_PEAK_TFLOPS,
                                     # the ceiling this path can reach: contraction is off by contract, so
                                     # every f64 instruction is one flop (39.3 = 78.6 / 2)
                                     "peak_no_fma": F64_VALU_NOFMA_TFLOPS, "frac_no_fma": tf / F64_VALU_NOFMA_TFLOPS,
                                     "flops_note": "wave-level instruction counts x 64 lanes (EXEC masks not applied): "
                                                   "an upper bound on the useful flops",
                                     "f64_flops_per_restart": fl, "valu_busy": pmc.get("valu_busy"),
                                     "source": os.path.relpath(PMC_FILE, ROOT) + ": SQ_INSTS_VALU_*_F64 x 64 lanes "
                                     "per restart and SQ_ACTIVE_INST_VALU per kernel of this command"}
            else:
                roof["secondary"] = None
            eng_bytes = None
            if pmc and pmc.get("kernels") and not T:
                eng_bytes = sum((2.0 * k_["FETCH_SIZE_kb_per_launch"] + k_["WRITE_SIZE_kb_per_launch"]) * 1024.0 * k_["launches"]
                                for k_ in pmc["kernels"].values() if "FETCH_SIZE_kb_per_launch" in k_) / total
            roof.update({"frac_path": None if T else total / elapsed * (16 * n + 16) / 1e9 / HBM_PEAK_GBS,
                         "bytes_per_restart_pmc": eng_bytes,
                         "traffic_ratio": (eng_bytes / (16 * n + 16)) if eng_bytes else None,
                         "valu_frac_no_fma": roof["secondary"]["frac_no_fma"] if roof["secondary"] else None,
                         "valu_active_lane_frac": (pmc or {}).get("valu_active_lane_frac")})
            info = {"grid": None, "block": 128, "lds_bytes": 0}
        else:
            kernel_ms, launches = hc.timing_mean()
            info = hc.last_launch()
            per_launch = cols * (K if pooled_kernel else 1)
            achieved = out_bytes * per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
            kname = "ik_quad_kernel"
            if kname == "ik_quad_kernel" and info["lds_bytes"] > 30000:
                kname = "ik_lane_kernel"  # one restart per lane (ik_lane64.hpp): 39 KB of LDS per single-wave workgroup
            wide_hbm = n > 8 and nat.get_option("wide_form") == 1
            if n > 8:  # the general solver: a restart per wave in LDS, or per lane in an HBM workspace
                kname = "wide_solve_kernel" if wide_hbm else "wide_solve_coop_kernel"
            kp = (pmc or {}).get("kernel_path")
            traffic, traffic_note, secondary = None, f"no PMC pass of this command under profiles/ (key: {key})", None
            if kp:
                traffic = kp["hbm_bytes_per_restart"] * per_launch
                traffic_note = (f"{os.path.relpath(PMC_FILE, ROOT)}: separate FETCH_SIZE / WRITE_SIZE passes of this command "
                                f"(2 x FETCH + WRITE), the {kp['launches']} launches of the timed repetitions: "
                                f"{kp['hbm_bytes_per_restart']:.0f} B per restart at the fabric counters against "
                                f"{out_bytes} B of outputs -- the difference is the kernel's few register spills (scratch: "
                                "per-wave private memory) and the targets / launch parameters it re-reads, not restart state")
                if wide_hbm:
                    # the general solver's HBM form streams a restart's whole SLSQP state through its workspace: the fabric
                    # bytes ARE the path's traffic, and they -- not the 8n + 24 output bytes -- are what it is bound by
                    gbps = kp["hbm_bytes_per_restart"] * total / elapsed / 1e9
                    traffic_note = (f"{os.path.relpath(PMC_FILE, ROOT)}: separate FETCH_SIZE / WRITE_SIZE passes of this command "
                                    f"(2 x FETCH + WRITE), the {kp['launches']} launches of the timed repetitions: "
                                    f"{kp['hbm_bytes_per_restart'] / 1e6:.2f} MB per restart at the fabric counters -- the restart "
                                    f"state of the general solver's HBM form lives in an HBM workspace (DESIGN.md section 5.6): "
                                    f"{gbps:.0f} GB/s = {gbps / HBM_PEAK_GBS:.2f} of the HBM peak at this line's rate")
                tf = total / elapsed * kp["f64_flops_per_restart"] / 1e12
                secondary = {"bound": "valu_f64", "achieved": tf, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": tf / F64_VALU_PEAK_TFLOPS, "peak_no_fma": F64_VALU_NOFMA_TFLOPS,
                             "frac_no_fma": tf / F64_VALU_NOFMA_TFLOPS,
                             "flops_note": ("wave-level instruction counts x 64 lanes (EXEC masks not applied): an upper bound on "
                                            "the useful flops; valu_active_lane_frac is the part of it that is lane work"
                                            + ("" if kname == "ik_lane_kernel" else "; the four lanes of a quad repeat the scalar parts")),
                             "f64_flops_per_restart": kp["f64_flops_per_restart"], "valu_busy": kp.get("valu_busy"),
                             "source": os.path.relpath(PMC_FILE, ROOT)}
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note, "kernel": kname,
                    # the whole path by SURVEY 8d's own unit (16n + 16 bytes per restart, seeds counted as read), what
                    # the fabric counters saw per restart and its ratio to that unit, and the f64 VALU view: fraction of
                    # the no-FMA ceiling at wave level and the part of it that is lane work (EXEC masks applied)
                    "frac_path": total / elapsed * (16 * n + 16) / 1e9 / HBM_PEAK_GBS,
                    "bytes_per_restart_pmc": kp["hbm_bytes_per_restart"] if kp else None,
                    "traffic_ratio": (kp["hbm_bytes_per_restart"] / (16 * n + 16)) if kp else None,
                    "valu_frac_no_fma": secondary["frac_no_fma"] if secondary else None,
                    "valu_active_lane_frac": kp.get("valu_active_lane_frac") if kp else None,
                    "kernel_ms": kernel_ms, "launches_timed": launches,
                    "algorithmic_bytes_per_launch": out_bytes * per_launch,
                    "algorithmic_bytes_per_unit": out_bytes, "unit_name": "restart (seeds are generated in-kernel: "
                    "the outputs are the whole per-restart traffic of this path)", "units_per_launch": per_launch,
                    "secondary": secondary}
        if T:
            metric = f"ik() calls/sec ({args.robot}, {T} targets x {R} restarts per step, 1e-6 tol; BASELINE.json config 5)"
            unit = "ik calls/s"
            workload = (f"{args.robot} {n}-DoF, {T} independent targets per step cut into one part per GPU, restart "
                        f"indices 0..{R - 1} each, SolutionMode::Speed with early exit (Robot::ik semantics), no collective")
        else:
            metric = ("random-restart IK solves/sec (Panda 7-DoF, 1e-6 tol)" if args.robot == "panda"
                      else f"random-restart IK solves/sec ({args.robot}, 1e-6 tol)")
            unit = "restarts/s"
            per = f"{R} random restarts per GPU per step" if args.scaling == "weak" else \
                f"{R} random restarts per step cut into one contiguous range per GPU"
            workload = (f"{args.robot} {n}-DoF, {per}, one target per step, SolutionMode::{mode.capitalize()}, "
                        "every restart run to termination")
        line = {
            "metric": metric,
            "value": total / elapsed,
            "unit": unit,
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "command_key": key,
                       "world": world, "backend": backend_name, "rank_devices": rank_devices,
                       # what the process group executed in every timed repetition (None: no process group)
                       "collectives": (["barrier", "all_reduce MIN int64 (key)", "all_reduce MIN int64 (index)",
                                        "all_reduce SUM f64 (winner x)", "all_reduce SUM f64 (winner f)",
                                        "all_reduce MAX f64 (elapsed)"] if distributed and not T else
                                       (["barrier", "all_reduce MAX f64 (elapsed)"] if distributed else None)),
                       "winner_f_last_step": (float(win_xf["f"].reshape(-1)[-1].item()) if win_xf else None),
                       "reps": len(rep_elapsed), "rep_reported": "median",
                       "value_reps": [total / e for e in rep_elapsed],
                       "value_min": total / max(rep_elapsed), "value_max": total / min(rep_elapsed),
                       "path": args.path, "restarts_per_gpu": cols, "tol_f": 1e-6, "solution_mode": mode,
                       "options_set": args.set_option or None,
                       "parallelism": (f"targets x{world}" if T else f"restart-range x{world}"),
                       "success_rate_last_step": (n_success / cols) if n_success is not None else None,
                       "solved_targets": solved_targets, "targets_per_step": per_step,
                       # NLopt's evaluation count per restart (what the reference's callback would be
                       # called), and the evaluations the kernels actually execute: NLopt re-evaluates an
                       # accepted line-search point that was not the first trial, the kernels do not
                       "mean_nlopt_evals_per_restart": mean_evals,
                       "mean_executed_evals_per_restart": mean_exec,
                       "executed_objective_gradient_evals_per_s":
                           (total / elapsed * mean_exec) if mean_exec else None,
                       # global winner of every timed step (first target of the step): the same for
                       # any number of ranks covering the same restart range
                       "winner_index_per_step": [int(v) for v in winners[:, 0].cpu().tolist()][:64],
                       "grid": info["grid"], "block": info["block"], "lds_bytes": info["lds_bytes"]},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline and not T:
            tables = robot.chain_tables()
            line["cpu_baseline"] = cpu_baseline(args.robot, tables, targets[W].cpu().numpy(),
                                                x0_host[W], mode, args.cpu_seconds)
            line["cpu_baseline"]["gpu_winner_same_target"] = int(winners[0, 0].item())
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

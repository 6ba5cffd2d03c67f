#!/usr/bin/env python3
"""Headline benchmark: random-restart IK solves/sec (Panda 7-DoF, tol_f = 1e-6).

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch: R = 65 536 restarts (BASELINE.json config 2: Panda
7-DoF, SolutionMode::Speed) of one synthetic reachable target, every restart run to termination (no early
exit), followed by the winner selection.  Targets, seeds x0 and all output buffers are resident in HBM before
the timed region.  The K timed steps of a run are the K targets of ONE launch of the solve kernel (each step
its own target, seed and output columns; at this size the lane-per-restart form, ik_lane64.hpp): the
persistent waves pull (step, restart) work items from one queue, so a run pays one fill and one drain of the
chip -- `config.workload` says so, and `config.other_configs.config2_single_launch` is the ISOLATED launch of
one step's 65 536 restarts.  The timed run is repeated until ~2 s of GPU work are inside the timed regions
(--reps overrides); `value` / `ms_per_step` are the MEDIAN repetition's.

Multi-GPU.  `python bench.py --gpus N` with no launcher around it starts the N ranks itself: it
re-executes under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1` with the same arguments (one process per GPU, RCCL over xGMI); under an outer launcher
(RANK / WORLD_SIZE in the environment: what the driver uses for N > 1) it is one of the ranks.  The
line's "n_gpus" is the number of ranks that actually joined the process group, and config carries
the world size, the backend string, RCCL's version, every rank's device (index, name, UUID: two ranks on
one physical GPU are refused unless OPTIK_BENCH_ONE_DEVICE=1) and the measured latency of each collective;
a --gpus that disagrees with WORLD_SIZE is refused.  With N > 1 the same invocation also times BASELINE
config 4 (4 M restarts, Quality, strong scaling) and config 5 (4096 targets x 256, cut per rank) in short
runs: `config.other_configs`, so that one scaling sweep yields all three curves.  `--inprocess` drives the
other multi-GPU form instead: ONE process, N devices behind the C ABI (optik_robot_set_devices: a host thread
per device, host min over N 16-byte records), through Robot.ik with SolutionMode::Quality and a restart
budget of R x N.
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
                    into one contiguous part per rank, no collective; value = ik() calls/s.  Runs the
                    product's round scheduling (128 indices restart-major, then the unsolved rest)

The JSON line carries:
  roofline      algorithmic HBM bytes of the dominant kernel / its mean duration (HIP events attached to
                that kernel's dispatches inside the C ABI); `traffic` = PMC bytes per launch when profiles/
                holds a PMC pass of this exact command, else null; `secondary` = the f64 vector-ALU view
                (HBM is not what binds this path): `frac_algorithmic` prices the oracle-counted f64
                operations of a restart (oracle/optik_oracle_flops.cpp) at this line's rate against the
                no-FMA ceiling, the PMC instruction counts sit beside it
  cpu_baseline  the CPU oracle (a port of the reference algorithm, NOT the reference binary) timed on this
                host's cores (1 thread = BASELINE config 1, half, all) on a bounded sample; `single_ik_ms` (one ik()
                call: 1 thread, and all cores under the reference's find_any rule) and `config5_ik_calls_per_s` on the
                targets of the GPU legs of the same names.  Runs BETWEEN the GPU legs (headline first, other configs last)
  config.other_configs   (N = 1, the default command) BASELINE configs 3, 4's one-GPU shard and 5, the isolated
                config-2 launch and a single ik() through the C ABI, each in a short run of its own
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
PMC_FILES = [os.path.join(ROOT, "profiles", f"r{r}_pmc_by_command.json") for r in (6, 5, 4)]
PMC_FILE = PMC_FILES[0]
ROBOT_SPECS = {"panda": ("panda.urdf", "panda_link0", "panda_link8"),
               "ur10": ("ur10.urdf", "base_link", "ee_link"),
               # a synthetic 8-DoF chain (tests/golden/robots/): the largest n the tuned kernels are built for
               "arm8": (os.path.join("..", "..", "tests", "golden", "robots", "arm8.urdf"), "l0", "l9"),
               # synthetic 10- and 16-joint chains: the general kernels of ik_wide.hpp
               "arm10": (os.path.join("..", "..", "tests", "golden", "robots", "arm10.urdf"), "l0", "l11"),
               "arm16": (os.path.join("..", "..", "tests", "golden", "robots", "arm16.urdf"), "l0", "l17")}


def command_key(args, world):
    """Identifies the workload a PMC pass was taken on (tools/profile_kernel_path.sh / pmc_kernel_path.py write the same key)."""
    return (f"robot={args.robot},restarts={args.restarts},steps={args.steps},warmup={args.warmup},"
            f"mode={args.mode},scaling={args.scaling},targets={args.targets},path=kernel,gpus={world}")


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
    spec = ROBOT_SPECS[robot]
    return Robot.from_urdf_file(os.path.join(ROOT, "optik_amd", "robots", spec[0]), spec[1], spec[2])


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


def cpu_baseline(robot_name, chain_tables, target7, x0, mode, seconds_budget=15.0, user_inputs=None):
    """Times the CPU oracle (checker code, used here only as the reported baseline) at 1 thread (BASELINE config 1),
    half the usable cores (the reference's advice, README.md:90-91) and all of them, and counts the f64 operations
    of a restart with the oracle's counting build (the algorithmic flops of the secondary roofline).
    With `user_inputs` (the targets of the GPU legs `single_ik` and `config5_all_4096_targets`) also the two figures a user
    of the reference would ask for (examples/example.rs:16-42): the time of ONE ik() call -- 1 thread, and all cores under
    the reference's own multi-thread rule (find_any, lib.rs:409-412) on persistent workers -- and ik() calls/s over
    config 5's 4 096 targets, one call per target on all cores."""
    from oracle import binding as ob
    ch = ob.make_chain(**chain_tables)
    cfg = ob.make_config(solution_mode=mode, tol_f=1e-6)
    flops = None
    try:
        ob.use_flops_build()
        n_f = 1500
        ob.flop_reset()
        ob.ik(ch, cfg, target7, x0, 0, n_f, n_threads=1, early_exit=False)
        c = ob.flop_counts()
        flops = {"per_restart": c["flops"] / n_f, "sample_restarts": n_f,
                 "by_op_per_restart": {k: c[k] / n_f for k in ("add_sub", "mul", "div", "sqrt")},
                 "not_counted_per_restart": {k: c[k] / n_f for k in ("compare", "sign_abs_minmax", "int_conversions")},
                 "source": "oracle/optik_oracle_flops.cpp: the oracle compiled with a counting double, restarts "
                           f"0..{n_f - 1} of the bench target, 1 thread (bit-identical results to the plain build)"}
    except Exception as e:  # noqa: BLE001 (no g++ on the host: the line says so instead of failing)
        flops = {"per_restart": None, "error": str(e)[:200]}
    flags = ob.use_native_build()  # -O3 -march=native of THIS host, same arithmetic
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
    user = cpu_user_figures(ob, ch, cores, user_inputs) if user_inputs is not None else {}
    return {"value": by_threads[str(cores)], "unit": "restarts/s", "cores": cores, "kind": "port", **user,
            # (BASELINE.json config 1 is the 1-thread figure; the reference advises half the cores)
            "value_1_thread": by_threads["1"], "value_half_cores": by_threads[str(max(1, cores // 2))],
            "by_threads": by_threads, "build": flags, "oracle_flops": flops,
            "sample": f"{robot_name}: the bench target, SolutionMode {mode}, every restart run to termination, threads "
                      f"pulling indices from a shared counter; " + "; ".join(sample),
            "winner": winner}


def cpu_user_figures(ob, ch, cores, inp):
    """The CPU oracle on the SAME inputs as the GPU legs `single_ik` (300 Robot.ik() calls, default SolverConfig) and
    `config5_all_4096_targets` (4 096 targets x at most 256 restarts, Speed, early exit).  ~1 s."""
    pose7, x0s = inp["pose7"], inp["x0s"]
    lo, hi = inp["single_ik_range"]
    dcfg = ob.make_config(solution_mode="speed")  # SolverConfig::default: max_time 0.1 s is never reached on these targets

    def calls(n_threads, rule):
        lat, solved = [], 0
        for t in range(lo, hi):
            t0 = time.perf_counter()
            r = ob.ik(ch, dcfg, pose7[t], x0s[t], 0, 1 << 20, n_threads=n_threads, early_exit=rule)
            lat.append(time.perf_counter() - t0)
            solved += bool(r["found"])
        lat.sort()
        return {"median_ms": lat[len(lat) // 2] * 1e3, "p90_ms": lat[int(len(lat) * 0.9)] * 1e3, "solved": solved,
                "calls": len(lat)}
    single = {"1_thread": calls(1, True)}
    if cores > 1:
        ob.pool_start(cores)
        try:
            single[f"{cores}_threads_find_any"] = calls(cores, "find_any")
        finally:
            ob.pool_stop()
    single["workload"] = ("the 300 targets of config.other_configs.single_ik, one ok_ik call each through ctypes: 1 thread (restarts in "
                          "index order, early exit), and all usable cores under the reference's find_any rule (lib.rs:409-412: the "
                          "first success in time wins, the others stop at their next objective call) on persistent worker threads")
    # config 5: one call per target (1 thread each, restart indices 0..255, early exit), the targets handed out to the
    # cores from a shared counter -- in C (ok_ik_many): sixteen Python threads calling through ctypes spend their time on
    # the interpreter lock, not on the solver
    T = len(pose7)
    c5cfg = ob.make_config(solution_mode="speed", max_restarts=256)
    # (the best of three passes: the first one also wakes the host's idle cores)
    found, dt = None, float("inf")
    for _ in range(3):
        found, _, d1 = ob.ik_many(ch, c5cfg, pose7, x0s, 256, cores)
        dt = min(dt, d1)
    return {"single_ik_ms": single,
            "config5_ik_calls_per_s": T / dt,
            "config5": {"targets": T, "seconds": dt, "solved": int(sum(bool(v) for v in found)), "threads": cores,
                        "workload": "the 4 096 targets of config.other_configs.config5_all_4096_targets (host-API leg), one single-threaded "
                                    "ok_ik per target (Speed, restart indices 0..255 in order, early exit), the targets handed out to the "
                                    "cores from a shared counter (ok_ik_many, in C)"}}


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


def device_uuid(index):
    try:
        return str(torch.cuda.get_device_properties(index).uuid)
    except Exception:  # noqa: BLE001 (older torch: no uuid attribute)
        return None


def check_distinct_devices(rank_devices, one_device):
    """Every rank of a multi-GPU line must sit on its own physical GPU.  The ranks of one node are told apart by their
    device index (LOCAL_RANK -> device): ranks that share one are refused.  The UUID torch reports is the second witness:
    equal UUIDs on distinct indices are reported (`config.uuid_warning`), not fatal -- a runtime that hands every device
    the same placeholder must not cost the node its scaling run.  OPTIK_BENCH_ONE_DEVICE=1 (tests on a one-GPU box)
    lifts the check.  Returns the warning or None."""
    world = len(rank_devices)
    if one_device or world < 2:
        return None
    idx = [d["device"] for d in rank_devices]
    if len(set(idx)) != world:
        raise SystemExit(f"ranks share a physical GPU (device indices {idx}): refusing to report {world} GPUs "
                         "(OPTIK_BENCH_ONE_DEVICE=1 allows it for tests)")
    uuids = [d.get("uuid") for d in rank_devices]
    if all(u is not None for u in uuids) and len(set(uuids)) != world:
        return f"distinct device indices {idx} but repeated UUIDs {uuids}"
    return None


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
    for _rep in range(max(1, args.reps or 5)):
        t0 = time.perf_counter()
        w_rep = run_steps(W, K)  # Robot.ik blocks until the G parts are done and reduced
        rep_elapsed.append(time.perf_counter() - t0)
        if winners is not None and winners != w_rep:
            raise SystemExit("winners differ between repetitions of the same steps")
        winners = w_rep
    elapsed = sorted(rep_elapsed)[len(rep_elapsed) // 2]
    total = float(R) * G * K
    out_bytes = 8 * n + 8 + 8 + 4 + 4
    parts = sorted(set(parts_used[-K:]))
    # (a call whose range is worth cutting must have been cut over every configured device: a line that says G GPUs
    # while some call ran on fewer would be a wrong scaling point)
    if R * G >= 65536 * G and parts != [G] and not os.environ.get("OPTIK_BENCH_ALLOW_UNCUT"):
        raise SystemExit(f"--inprocess --gpus {G}: the timed calls were cut over {parts} device(s), not {G}")
    line = {
        "metric": "random-restart IK solves/sec (Panda 7-DoF, 1e-6 tol)" if args.robot == "panda"
                  else f"random-restart IK solves/sec ({args.robot}, 1e-6 tol)",
        "value": total / elapsed, "unit": "restarts/s", "n_gpus": min(parts_used[-K:]) if parts_used else 0, "steps": K, "warmup": W,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.robot} {n}-DoF, {R} random restarts per GPU per step (one Robot.ik call over "
                               f"[0, {R * G})), one target per step, SolutionMode::Quality, every restart run to termination",
                   "inprocess": True, "devices": devices, "devices_configured": robot.num_devices(),
                   "device_uuids": [device_uuid(d) for d in devices],
                   # (n_gpus above is what the calls were really cut over -- the host API keeps a range that is not worth
                   # cutting on one device -- not what was configured)
                   "parts_per_call": parts, "world": 1,
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


class Ctx:
    """What every workload of one invocation shares: the device, the process group, the ranks."""

    def __init__(self, dev, rank, world, distributed, dist, backend_name):
        self.dev, self.rank, self.world = dev, rank, world
        self.distributed, self.dist, self.backend_name = distributed, dist, backend_name
        self.robots = {}

    def robot(self, name):
        if name not in self.robots:
            r = load_chain(name)
            self.robots[name] = (r, r.hip_chain(self.dev))
        return self.robots[name]

    def fence(self):
        """barrier + synchronize on both sides of a timed region (the contract's bracket)."""
        torch.cuda.synchronize()
        if self.distributed:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if not self.distributed:
            return seconds
        from optik_amd.parallel import _all_reduce
        t = torch.tensor([seconds], dtype=torch.float64, device=self.dev)
        _all_reduce(t, self.dist.ReduceOp.MAX)
        return float(t.item())


def run_workload(ctx, robot_name, mode, scaling, T, R, K, W, reps, find_any=False, min_timed_s=0.0):
    """One workload of the bench: W warm-up steps, then `reps` repetitions of K timed steps, each repetition
    bracketed by barrier + synchronize and reduced to the MAX over ranks.  reps = None: as many as put
    `min_timed_s` of work inside the timed regions (at least 5)."""
    from optik_amd import _native as nat
    from optik_amd.parallel import I64_MAX, shard_range, select_winner, gather_winner_x
    dev, rank, world, distributed = ctx.dev, ctx.rank, ctx.world, ctx.distributed
    robot, hc = ctx.robot(robot_name)
    n = robot.num_positions()
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
        begin, end = shard_range(0, R * world if scaling == "weak" else R, rank, world)
        cols = end - begin
    if T_loc < 1 or cols < 1:
        raise SystemExit("more ranks than work items: nothing to do on this rank")
    # the steps of a run are the targets of ONE launch of the solve kernel (each step its own target, seed and
    # output columns) -- the persistent waves pull (step, restart) items from one queue, so the run pays one
    # fill and one drain of the chip instead of one per step
    pooled = not T
    # (one set of output buffers per run length: the timed run of K steps and the warm-up run of W)
    kbufs = {c: hc.alloc_ik_buffers(c, cols) for c in {K, W} if c} if pooled else {}
    win_xf = {}  # the global winners' x and f of the last run (every rank holds them, as Robot::ik returns them)

    def exchange(rec):
        """The cross-rank half of a step (lib.rs:397-413 over the ranks): two 8-byte min-all-reduces pick the
        winner, a sum-all-reduce of <= 64 B per target hands its x and f to every rank."""
        win = select_winner(rec, mode, distributed)
        if distributed:
            win_xf["x"], win_xf["f"] = gather_winner_x(rec, win, begin, end, True)
        return win
    # plumbing first-use costs (torch's lazily loaded elementwise kernels, the communicator of the
    # first collective) are paid here on dummy records, not inside the timed region when W = 0
    select_winner({"win_idx": torch.zeros(1, dtype=torch.int64, device=dev),
                   "win_key": torch.zeros(1, dtype=torch.float64, device=dev)}, mode, distributed and not T)
    torch.cuda.synchronize()

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
                              | (nat.IK_FIND_ANY if find_any else 0), per_restart=False)
            solved = out["win_idx"] >= 0
            win[live[solved]] = out["win_idx"][solved]
            live = live[~solved]
            b, rnd = e, rnd * 4
        return win

    def run_steps(first, count):
        """`count` steps starting at step `first`; returns the per-step winners ([count, T_loc])."""
        if pooled:
            i = first + t_lo
            kb = kbufs[count]
            hc.ik_batch(cfg, targets[i:i + count], x0[i:i + count], begin, end, bufs=kb, per_restart=True)
            return exchange(kb).clone().reshape(count, 1)
        winners = []
        for k in range(count):
            i = (first + k) * per_step + t_lo
            if mode == "speed":
                winners.append(batch_rounds(targets[i:i + T_loc], x0[i:i + T_loc]))
                continue
            out = hc.ik_batch(cfg, targets[i:i + T_loc], x0[i:i + T_loc], begin, end, per_restart=False)
            winners.append(select_winner(out, mode, False).clone())
        return torch.stack(winners)

    if W:
        run_steps(0, W)
    ctx.fence()
    rep_elapsed, winners = [], None
    hc.set_timing(True)  # (HIP events around the dominant kernel's launches of ALL the timed repetitions)
    n_reps = reps
    while True:
        ctx.fence()
        t0 = time.perf_counter()
        w_rep = run_steps(W, K)
        ctx.fence()
        el = ctx.max_over_ranks(time.perf_counter() - t0)
        rep_elapsed.append(el)
        if winners is not None and not torch.equal(winners, w_rep) and not (T and find_any):
            raise SystemExit("winners differ between repetitions of the same steps")
        winners = w_rep
        if n_reps is None:
            # (every rank derives the count from the same MAX-reduced time: they stay in step)
            n_reps = int(min(200, max(5, np.ceil(min_timed_s / max(el, 1e-6)))))
        if len(rep_elapsed) >= n_reps:
            break
    elapsed = sorted(rep_elapsed)[len(rep_elapsed) // 2]  # the median repetition (upper median for even counts)
    kernel_ms, launches = hc.timing_mean()
    info = hc.last_launch()
    hc.set_timing(False)
    res = dict(robot=robot, hc=hc, n=n, elapsed=elapsed, rep_elapsed=rep_elapsed, winners=winners, cols=cols,
               T_loc=T_loc, begin=begin, end=end, per_step=per_step, pooled=pooled, kernel_ms=kernel_ms,
               launches=launches, info=info, targets=targets, x0_host=x0_host, win_xf=win_xf,
               units_per_step=float(T) if T else float(cols) * world)
    res["value"] = res["units_per_step"] * K / elapsed
    res["solved_targets"] = int((winners >= 0).logical_and(winners < torch.iinfo(torch.int64).max).sum().item())
    if pooled:
        last = dict(status=kbufs[K]["status"][(K - 1) * cols:K * cols], evals=kbufs[K]["evals"])
        res["n_success"] = int((last["status"] == nat.RES_STOPVAL).sum().item())
        res["mean_evals"] = float(last["evals"].double().mean().item())
    else:
        res["n_success"], res["mean_evals"] = None, None
    return res


def user_inputs(ctx):
    """The targets of the two user-facing legs (a single ik(), config 5 through the host API): 4 096 reachable Panda poses
    FK(q), q and the seeds uniform in the limits -- as pose7 (for the CPU oracle) and as 4x4 matrices (for Robot.ik)."""
    robot, hc = ctx.robot("panda")
    rng = np.random.default_rng(5)
    lb, ub = (np.array(v) for v in robot.joint_limits())
    T = 4096
    pose = hc.fk_batch(torch.tensor(rng.uniform(lb, ub, size=(T, 7)).T.copy(), device=ctx.dev)).T.cpu().numpy()
    i, j, k, w = pose[:, 3], pose[:, 4], pose[:, 5], pose[:, 6]
    m = np.zeros((T, 4, 4))
    m[:, 0, 0] = w*w+i*i-j*j-k*k; m[:, 0, 1] = 2*(i*j-w*k); m[:, 0, 2] = 2*(w*j+i*k)
    m[:, 1, 0] = 2*(w*k+i*j); m[:, 1, 1] = w*w-i*i+j*j-k*k; m[:, 1, 2] = 2*(j*k-w*i)
    m[:, 2, 0] = 2*(i*k-w*j); m[:, 2, 1] = 2*(w*i+j*k); m[:, 2, 2] = w*w-i*i-j*j+k*k
    m[:, :3, 3] = pose[:, :3]
    m[:, 3, 3] = 1.0
    return {"pose7": pose, "m": m, "x0s": rng.uniform(lb, ub, size=(T, 7)), "single_ik_range": (64, 64 + 300)}


def other_configs_one_gpu(ctx, args, primary, user):
    """N = 1, the default command: the rest of BASELINE.json's configurations and the latency figures, each in a
    short run of its own (about ten seconds in all), so that the driver's ONE line carries the whole contract."""
    from optik_amd import SolverConfig
    from optik_amd import _native as nat
    out = {}
    dev = ctx.dev

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2], ts

    # config 2 as BASELINE words it: ONE launch of 65 536 Panda restarts (fill + drain of the chip included)
    robot, hc = ctx.robot("panda")
    tg, x0t = primary["targets"], torch.tensor(primary["x0_host"], device=dev)
    W = args.warmup
    cfg = nat.make_config(solution_mode="speed", tol_f=1e-6)
    b1 = hc.alloc_ik_buffers(1, 65536)
    med, ts = timed(lambda: hc.ik_batch(cfg, tg[W:W + 1], x0t[W:W + 1], 0, 65536, bufs=b1), 9)
    out["config2_single_launch"] = {"workload": "Panda, ONE launch of 65 536 restarts of one target (Speed, every restart to "
                                                "termination) + selection, synchronised before and after",
                                    "single_launch_ms": med * 1e3, "restarts_per_s": 65536 / med,
                                    "ms_all": [t * 1e3 for t in ts]}
    del b1
    # config 3: UR10, one launch of 2^20 restarts, tol_f 1e-6 and 1e-12 (tests/test_ik.rs:99), FK(x) == T checked
    r3 = run_workload(ctx, "ur10", "quality", "weak", 0, 1 << 20, 1, 1, 3)
    urobot, uhc = ctx.robot("ur10")
    c3 = {"workload": "UR10 6-DoF, one launch of 2^20 restarts of one target, SolutionMode::Quality, every restart to termination",
          "tol_f_1e-6": {"restarts_per_s": r3["value"], "ms": r3["elapsed"] * 1e3, "kernel_ms": r3["kernel_ms"],
                         "success_rate": r3["n_success"] / r3["cols"], "winner": int(r3["winners"][0, 0].item())}}
    cfg12 = nat.make_config(solution_mode="quality", tol_f=1e-12)
    b3 = uhc.alloc_ik_buffers(1, 1 << 20)
    tg3, x03 = r3["targets"][1:2], torch.tensor(r3["x0_host"][1:2], device=dev)
    med, _ = timed(lambda: uhc.ik_batch(cfg12, tg3, x03, 0, 1 << 20, bufs=b3), 3)
    ok = b3["status"] == nat.RES_STOPVAL
    wx = b3["win_x"][0].view(-1, 1).contiguous()
    pose = uhc.fk_batch(wx)[:, 0]
    t7 = tg3[0]
    err = float(torch.maximum((pose[:3] - t7[:3]).abs().max(),
                              torch.minimum((pose[3:] - t7[3:]).abs().max(), (pose[3:] + t7[3:]).abs().max())).item())
    c3["tol_f_1e-12"] = {"restarts_per_s": (1 << 20) / med, "ms": med * 1e3, "success_rate": float(ok.double().mean().item()),
                         "winner": int(b3["win_idx"][0].item()), "winner_pose_error": err, "winner_pose_ok_1e-6": err < 1e-6}
    out["config3_ur10_1M"] = c3
    del b3
    # config 4: one GPU's shard of the 4 M (524 288 restarts, Quality)
    r4 = run_workload(ctx, "panda", "quality", "weak", 0, 524288, 1, 1, 3)
    out["config4_one_gpu_shard"] = {"workload": "Panda, 524 288 restarts (one GPU's contiguous eighth of BASELINE config 4's 4 M), "
                                                "SolutionMode::Quality, one launch + selection",
                                    "restarts_per_s": r4["value"], "ms": r4["elapsed"] * 1e3, "kernel_ms": r4["kernel_ms"],
                                    "winner": int(r4["winners"][0, 0].item())}
    # config 5: 4096 targets x 256 through the product's round scheduling; one GPU's share is 512 targets
    for name, T in (("config5_one_gpu_share_512_targets", 512), ("config5_all_4096_targets", 4096)):
        r5 = run_workload(ctx, "panda", "speed", "weak", T, 256, 2, 1, 3, find_any=False)
        r5f = run_workload(ctx, "panda", "speed", "weak", T, 256, 2, 1, 3, find_any=True)
        out[name] = {"workload": f"Panda, {T} independent targets x restart indices 0..255, Speed with early exit, rounds of 128 "
                                 "restart-major then the unsolved rest (the kernel-layer form of robot_host.cpp:ik_batch_on_device)",
                     "ik_calls_per_s_deterministic": r5["value"], "ik_calls_per_s_find_any": r5f["value"],
                     "ms_per_batch_deterministic": r5["elapsed"] / 2 * 1e3, "ms_per_batch_find_any": r5f["elapsed"] / 2 * 1e3,
                     "solved_targets_of": [r5["solved_targets"], 2 * T]}
    # ... and the same 4096 through the host API (numpy in, numpy out: PCIe and the 4x4 -> pose conversion included)
    T = 4096
    m, x0s = user["m"], user["x0s"]
    scfg = SolverConfig(solution_mode="speed", max_time=0.0, max_restarts=256)
    med, _ = timed(lambda: robot.ik_batch_arrays(scfg, m, x0s), 5)
    out["config5_all_4096_targets"]["ik_calls_per_s_host_api"] = T / med
    # a single ik() through the C ABI (the reference's own timing loop: examples/example.rs:16-42), default SolverConfig
    lat, n_solved = [], 0
    dcfg = SolverConfig()
    for t in range(user["single_ik_range"][0]):
        robot.ik(dcfg, m[t], x0s[t].tolist())
    for t in range(*user["single_ik_range"]):
        t0 = time.perf_counter()
        r = robot.ik(dcfg, m[t], x0s[t].tolist())
        lat.append(time.perf_counter() - t0)
        n_solved += r is not None
    lat.sort()
    out["single_ik"] = {"workload": "Robot.ik(SolverConfig()) on one reachable Panda target per call through optik_robot_ik_ex "
                                    "(ctypes), the reference's default first-success rule, 300 calls",
                        "solved": n_solved, "median_ms": lat[len(lat) // 2] * 1e3, "p90_ms": lat[int(len(lat) * 0.9)] * 1e3,
                        "min_ms": lat[0] * 1e3}
    return out


def other_configs_multi_gpu(ctx, args):
    """N > 1: BASELINE config 4 (4 M restarts, Quality, cut per rank) and config 5 (4096 targets x 256, cut per rank)
    in short runs of the same invocation -- one scaling sweep of the driver yields all three curves."""
    r4 = run_workload(ctx, "panda", "quality", "strong", 0, 1 << 22, 2, 1, 3)
    r5 = run_workload(ctx, "panda", "speed", "weak", 4096, 256, 4, 1, 3)
    if ctx.rank != 0:
        return None
    return {"config4_strong_quality": {"workload": f"Panda, 4 194 304 restarts per step cut into one contiguous range per GPU "
                                                   f"(x{ctx.world}), SolutionMode::Quality, two min-all-reduces + winner broadcast per step",
                                       "restarts_per_s": r4["value"], "ms_per_step": r4["elapsed"] / 2 * 1e3,
                                       "restarts_per_gpu": r4["cols"], "winner_index_per_step": [int(v) for v in r4["winners"][:, 0].cpu().tolist()],
                                       "scaling": "strong"},
            "config5_targets": {"workload": f"Panda, 4096 targets x 256 restart indices per step, targets cut into one part per GPU "
                                            f"(x{ctx.world}), Speed with early exit, no collective",
                                "ik_calls_per_s": r5["value"], "ms_per_step": r5["elapsed"] / 4 * 1e3,
                                "targets_per_gpu": r5["T_loc"], "scaling": "strong"}}


def collective_latency(ctx):
    """Microseconds per collective of a step's exchange, measured after the timed runs (20 of each, synchronised)."""
    if not ctx.distributed:
        return None
    from optik_amd.parallel import _all_reduce
    dist, dev = ctx.dist, ctx.dev
    out = {}
    cases = {"all_reduce_min_int64_8B": (torch.zeros(1, dtype=torch.int64, device=dev), dist.ReduceOp.MIN),
             "all_reduce_sum_f64_64B": (torch.zeros(8, dtype=torch.float64, device=dev), dist.ReduceOp.SUM)}
    for name, (t, op) in cases.items():
        _all_reduce(t, op)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            _all_reduce(t, op)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / 20 * 1e6
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dist.barrier()
    out["barrier"] = (time.perf_counter() - t0) / 20 * 1e6
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs (= ranks) of this node; without a launcher N > 1 ranks are spawned here")
    ap.add_argument("--inprocess", action="store_true",
                    help="one process driving --gpus devices through optik_robot_set_devices (Robot.ik, "
                         "Quality, restart budget R x N) instead of one process per GPU")
    ap.add_argument("--reps", type=int, default=None,
                    help="repetitions of the K-step timed run: value / ms_per_step are the MEDIAN repetition's, "
                         "config.value_reps lists all of them; default: as many as put ~2 s inside the timed regions (>= 5)")
    ap.add_argument("--steps", type=int, default=48,
                    help="timed steps of a run: the targets of ONE launch of the solve kernel")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--restarts", type=int, default=None,
                    help="restarts per step: per GPU (weak) or in total (strong); per target with --targets; "
                         "default 65536 (256 with --targets)")
    ap.add_argument("--robot", default="panda", choices=sorted(ROBOT_SPECS))
    ap.add_argument("--mode", default="speed", choices=["speed", "quality"], help="SolutionMode (config.rs:3-8)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--targets", type=int, default=0,
                    help="config 5: targets per step (cut into one part per rank), each an independent ik() call")
    ap.add_argument("--path", default="kernel", choices=["auto", "kernel"],
                    help="(kept for the profile scripts) the single-launch solvers: one persistent solve kernel for the "
                         "run's steps, restart state in registers and LDS -- the only path since the streaming engine "
                         "of rounds 1-4 was retired")
    ap.add_argument("--find-any", action="store_true",
                    help="--targets: the reference's default reading of should_exit (rayon find_any, lib.rs:409-412: ANY "
                         "success ends a target's other restarts) as Robot.ik_batch runs it unless set_parallelism(1); the "
                         "default here is the deterministic reading (lowest solved index), which every rank count reproduces")
    ap.add_argument("--force-distributed", action="store_true",
                    help="run the process-group path (init_process_group, the min / sum / max all-reduces, "
                         "all_gather_object) even with ONE rank: RCCL on a single GPU")
    ap.add_argument("--set-option", action="append", default=[], metavar="NAME=VALUE",
                    help="a tuning option of the kernel layer for this run (include/optik_hip.h: optik_hip_set_option), "
                         "e.g. solve_kernel=1; experiments only -- the line records what was set")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip config.other_configs (profiling runs: only the headline workload's kernels in the trace)")
    ap.add_argument("--cpu-seconds", type=float, default=13.0,
                    help="budget of the restarts/s legs of cpu_baseline (the user-facing legs add ~1 s)")
    args = ap.parse_args()
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
    one_device = os.environ.get("OPTIK_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_device else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but this node shows {torch.cuda.device_count()} device(s)")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    rccl_version = None
    me = {"rank": rank, "device": int(torch.cuda.current_device()), "name": torch.cuda.get_device_name(dev_index),
          "uuid": device_uuid(dev_index), "pid": os.getpid()}
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
            try:
                rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:  # noqa: BLE001
                rccl_version = None
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != world:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, WORLD_SIZE says {world}")
        # proof that N ranks joined, each on its own GPU: (rank, current device, device name, UUID, pid)
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, me)
        backend_name = str(dist.get_backend())
        uuid_warning = check_distinct_devices(rank_devices, one_device)
    else:
        rank_devices = [me]
        backend_name = None
        uuid_warning = None

    from optik_amd import _native as nat
    for kv in args.set_option:
        name, _, val = kv.partition("=")
        nat.set_option(name, int(val) if val.lstrip("-").isdigit() else val)

    ctx = Ctx(dev, rank, world, distributed, dist, backend_name)
    R, K, W, T, mode = args.restarts, args.steps, args.warmup, args.targets, args.mode
    primary = run_workload(ctx, args.robot, mode, args.scaling, T, R, K, W, args.reps, args.find_any, min_timed_s=2.0)
    # the other workloads of the same invocation (every rank takes part in the multi-rank ones)
    headline = args.robot == "panda" and not T and mode == "speed" and args.scaling == "weak" and R == 65536
    want_others = not args.no_other_configs and headline
    # Order of the legs at N = 1: GPU (the headline above), CPU (the oracle baseline), GPU (the other configurations) -- the
    # driver's 5-s GPU-busy sampler then meets GPU work at both ends of the run instead of ~11 s of CPU work at its end.
    user = None
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not T:
        if want_others and not args.force_distributed:
            try:
                user = user_inputs(ctx)
            except Exception:  # noqa: BLE001
                user = None
        # (a failure of the baseline leg -- no compiler on the host, say -- must not cost the headline line either)
        try:
            cpu = cpu_baseline(args.robot, primary["robot"].chain_tables(), primary["targets"][W].cpu().numpy(),
                               primary["x0_host"][W], mode, args.cpu_seconds, user)
            cpu["gpu_winner_same_target"] = int(primary["winners"][0, 0].item())
        except Exception as e:  # noqa: BLE001
            cpu = {"error": f"{type(e).__name__}: {e}"[:300], "kind": "port"}
    others = None
    if want_others:
        # (a failure here -- the same on every rank: same code, same sizes -- must not cost the headline line)
        try:
            if world > 1:
                others = other_configs_multi_gpu(ctx, args)
            elif not args.force_distributed:
                others = other_configs_one_gpu(ctx, args, primary, user or user_inputs(ctx))
        except Exception as e:  # noqa: BLE001
            others = {"error": f"{type(e).__name__}: {e}"[:300]}
    coll_us = collective_latency(ctx)

    if rank == 0:
        n, hc, robot = primary["n"], primary["hc"], primary["robot"]
        elapsed, rep_elapsed, winners, cols = primary["elapsed"], primary["rep_elapsed"], primary["winners"], primary["cols"]
        total = primary["units_per_step"] * K
        out_bytes = 8 * n + 8 + 8 + 4 + 4  # x[n] + f + key + status + evals written per restart
        key = command_key(args, world)
        pmc = pmc_for(key)
        kernel_ms, launches, info = primary["kernel_ms"], primary["launches"], primary["info"]
        per_launch = cols * (K if primary["pooled"] else 1)
        achieved = out_bytes * per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        kname = "ik_lane_kernel" if info["lds_bytes"] > 30000 else "ik_quad_kernel"  # (39 KB of LDS per single-wave workgroup)
        wide_hbm = n > 8 and nat.get_option("wide_form") == 1
        if n > 8:  # the general solver: a restart per wave in LDS, or per lane in an HBM workspace
            kname = "wide_solve_kernel" if wide_hbm else "wide_solve_coop_kernel"
        kp = (pmc or {}).get("kernel_path")
        traffic, traffic_note, secondary = None, f"no PMC pass of this command under profiles/ (key: {key})", None
        oflops = ((cpu or {}).get("oracle_flops") or {}).get("per_restart")
        if kp or oflops:
            secondary = {"bound": "valu_f64", "peak": F64_VALU_PEAK_TFLOPS, "peak_no_fma": F64_VALU_NOFMA_TFLOPS, "unit": "TFLOP/s"}
            if oflops:
                # the ALGORITHMIC view: the f64 operations the reference algorithm performs per restart, counted
                # by the oracle, at this line's rate -- a numerator that does not grow when the kernel wastes instructions
                tfa = total / elapsed * oflops / 1e12
                secondary.update({"achieved": tfa, "frac": tfa / F64_VALU_PEAK_TFLOPS,
                                  "frac_algorithmic": tfa / F64_VALU_NOFMA_TFLOPS,
                                  "f64_flops_per_restart_oracle": oflops,
                                  "flops_note": "achieved / frac_algorithmic: oracle-counted + - * / sqrt of a restart (cpu_baseline."
                                                "oracle_flops) x restarts/s, against the no-FMA ceiling (contraction is forbidden by "
                                                "the bit-exactness contract); frac is the same against the FMA peak"})
        if kp:
            traffic = kp["hbm_bytes_per_restart"] * per_launch
            traffic_note = (f"{os.path.relpath(PMC_FILE, ROOT)}: separate FETCH_SIZE / WRITE_SIZE passes of this command "
                            f"(2 x FETCH + WRITE), the {kp['launches']} launches of the timed repetitions: "
                            f"{kp['hbm_bytes_per_restart']:.0f} B per restart at the fabric counters against "
                            f"{out_bytes} B of outputs")
            if wide_hbm:
                gbps = kp["hbm_bytes_per_restart"] * total / elapsed / 1e9
                traffic_note += (f"; the restart state of the general solver's HBM form lives in an HBM workspace (DESIGN.md "
                                 f"section 5.6): {gbps:.0f} GB/s = {gbps / HBM_PEAK_GBS:.2f} of the HBM peak at this line's rate")
            tf = total / elapsed * kp["f64_flops_per_restart"] / 1e12
            secondary.update({"achieved_pmc": tf, "frac_no_fma": tf / F64_VALU_NOFMA_TFLOPS,
                              "f64_flops_per_restart": kp["f64_flops_per_restart"], "valu_busy": kp.get("valu_busy"),
                              "pmc_note": "wave-level instruction counts x 64 lanes (EXEC masks not applied): an upper bound on "
                                          "the useful flops; valu_active_lane_frac is the part of it that is lane work",
                              "source": os.path.relpath(PMC_FILE, ROOT)})
            if "achieved" not in secondary:
                secondary.update({"achieved": tf, "frac": tf / F64_VALU_PEAK_TFLOPS})
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note, "kernel": kname,
                # the whole path by SURVEY 8d's own unit (16n + 16 bytes per restart, seeds counted as read), what
                # the fabric counters saw per restart and its ratio to that unit, and the f64 VALU view: the oracle-counted
                # operations at this rate against the no-FMA ceiling, the wave-level PMC count and its lane-work part
                "frac_path": total / elapsed * (16 * n + 16) / 1e9 / HBM_PEAK_GBS,
                "bytes_per_restart_pmc": kp["hbm_bytes_per_restart"] if kp else None,
                "traffic_ratio": (kp["hbm_bytes_per_restart"] / (16 * n + 16)) if kp else None,
                "valu_frac_algorithmic": (secondary or {}).get("frac_algorithmic"),
                "valu_frac_no_fma": (secondary or {}).get("frac_no_fma"),
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
                        f"every restart run to termination; the {K} steps of a timed run are the {K} targets of ONE launch of "
                        f"the solve kernel ({kname}: {K * cols} restarts per launch per GPU -- one fill and one drain of the "
                        "chip per run; the isolated one-step launch is config.other_configs.config2_single_launch)")
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
                       "world": world, "backend": backend_name, "rccl_version": rccl_version, "rank_devices": rank_devices, "uuid_warning": uuid_warning,
                       # what the process group executed in every timed repetition (None: no process group)
                       "collectives": (["barrier", "all_reduce MIN int64 (key)", "all_reduce MIN int64 (index)",
                                        "all_reduce SUM f64 (winner x)", "all_reduce SUM f64 (winner f)",
                                        "all_reduce MAX f64 (elapsed)"] if distributed and not T else
                                       (["barrier", "all_reduce MAX f64 (elapsed)"] if distributed else None)),
                       "collective_us": coll_us,
                       "winner_f_last_step": (float(primary["win_xf"]["f"].reshape(-1)[-1].item()) if primary["win_xf"] else None),
                       "reps": len(rep_elapsed), "rep_reported": "median",
                       "timed_region_s": sum(rep_elapsed),
                       "value_reps": [total / e for e in rep_elapsed][:64],
                       "value_min": total / max(rep_elapsed), "value_max": total / min(rep_elapsed),
                       "path": "kernel", "restarts_per_gpu": cols, "tol_f": 1e-6, "solution_mode": mode,
                       "options_set": args.set_option or None,
                       "parallelism": (f"targets x{world}" if T else f"restart-range x{world}"),
                       "success_rate_last_step": (primary["n_success"] / cols) if primary["n_success"] is not None else None,
                       "solved_targets": primary["solved_targets"], "targets_per_step": primary["per_step"],
                       # NLopt's evaluation count per restart (what the reference's callback would be called)
                       "mean_nlopt_evals_per_restart": primary["mean_evals"],
                       # global winner of every timed step (first target of the step): the same for
                       # any number of ranks covering the same restart range
                       "winner_index_per_step": [int(v) for v in winners[:, 0].cpu().tolist()][:64],
                       "grid": info["grid"], "block": info["block"], "lds_bytes": info["lds_bytes"],
                       "other_configs": others},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if cpu and others is not None and "single_ik" in others and "single_ik_ms" in cpu:
            # the CPU figure beside each user-facing GPU figure (same targets): a GPU/CPU ratio is not credit -- one default
            # ik() call takes the MI355X ~2.6 x as long as ONE host thread, which is what section 8.3 of DESIGN.md says it must
            others["single_ik"]["cpu_median_ms"] = {k: v["median_ms"] for k, v in cpu["single_ik_ms"].items() if isinstance(v, dict)}
            others["config5_all_4096_targets"]["cpu_ik_calls_per_s"] = cpu["config5_ik_calls_per_s"]
        if cpu and "value_1_thread" in cpu and others is not None:
            others["config1_cpu_1_thread"] = {"workload": "Panda, the bench target, the CPU oracle on ONE host thread (BASELINE config 1: "
                                                          "plumbing, no GPU)", "restarts_per_s": cpu["value_1_thread"],
                                              "winner": cpu["winner"], "gpu_winner_same_target": cpu["gpu_winner_same_target"]}
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

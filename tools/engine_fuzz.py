#!/usr/bin/env python3
"""Scheduler fuzz: random jobs (robot, targets, restart ranges, mode, tolerances, weights) under
random engine options (pool size, sub-pools, NNLS budget / slack, tail hand-over, compaction) must
give, restart for restart, the bits of the single-kernel path (status, evaluations, x, f, winners).
Usage: python tools/engine_fuzz.py [rounds] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optik_amd import Robot  # noqa: E402
from optik_amd import _native as nat  # noqa: E402

ROBOTS = os.path.join(ROOT, "optik_amd", "robots")
SPECS = {"panda": ("panda.urdf", "panda_link0", "panda_link8"), "ur10": ("ur10.urdf", "base_link", "ee_link"),
         "panda_hand": ("panda.urdf", "panda_link0", "panda_hand"), "panda5": ("panda.urdf", "panda_link0", "panda_link5")}
# options of the kernel layer (include/optik_hip.h: optik_hip_set_option); None = the default
KNOBS = {"engine_slots": [None, 1024, 1000, 2560, 4096, 20000],
         "engine_pools": [None, 1, 2, 3, 4],
         "engine_nnls_budget": [None, 1, 2, 3, 6, 12],
         "engine_nnls_slack": [None, 0, 1, 2, 100],
         "engine_tail_max": [None, 0, 7, 300, 100000],
         "engine_compact": [None, None, 0],
         # the reference answers come from the single-launch path: the quad solver, or forced onto the lane-per-restart form
         "solve_kernel": [None, None, "lane64", "quad"]}

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    chains = {k: Robot.from_urdf_file(os.path.join(ROBOTS, f), b, e) for k, (f, b, e) in SPECS.items()}
    bad = 0
    for it in range(rounds):
        name = rng.choice(list(SPECS))
        rb = chains[name]
        hc = rb.hip_chain("cuda:0")
        lb, ub = (np.array(v) for v in rb.joint_limits())
        mode = rng.choice(["speed", "quality"])
        cfg = nat.make_config(solution_mode=mode, tol_f=float(rng.choice([1e-6, 1e-8, 1e-4])),
                              linear_weight=tuple(rng.choice([1.0, 0.5, 2.0], 3)),
                              angular_weight=tuple(rng.choice([1.0, 0.7, 1.3], 3)))
        early = mode == "speed" and rng.random() < 0.4
        flags = 1 if early else 0
        jobs = []
        for _ in range(int(rng.integers(1, 5))):
            T = int(rng.integers(1, 6))
            q = rng.uniform(lb, ub, size=(T, len(lb)))
            tg = hc.fk_batch(torch.tensor(q.T.copy(), device="cuda")).T.contiguous()
            x0 = torch.tensor(rng.uniform(lb, ub, size=(T, len(lb))), device="cuda")
            begin = int(rng.integers(0, 3000))
            jobs.append((tg, x0, begin, begin + int(rng.integers(1, 1500))))
        knobs = {k: v[int(rng.integers(len(v)))] for k, v in KNOBS.items()}
        knobs = {k: v for k, v in knobs.items() if v is not None}
        sk = knobs.pop("solve_kernel", None)
        with nat.options(**({"solve_kernel": sk} if sk else {})):
            ref = None if early else [hc.ik_batch(cfg, t, x, b, e) for t, x, b, e in jobs]
            torch.cuda.synchronize()
        with nat.options(**knobs):
            outs = [hc.engine_submit(cfg, t, x, b, e, flags=flags) for t, x, b, e in jobs]
            hc.engine_run()
            torch.cuda.synchronize()
        if early:  # compare with the engine under default options: early exit is order dependent only in what it skips
            ref = [hc.engine_submit(cfg, t, x, b, e, flags=flags) for t, x, b, e in jobs]
            hc.engine_run()
            torch.cuda.synchronize()
        if sk:
            knobs["solve_kernel(reference)"] = sk
        ok = True
        for r, o in zip(ref, outs):
            if early:
                ok = ok and torch.equal(r["win_idx"], o["win_idx"]) and torch.equal(r["win_x"].view(torch.int64), o["win_x"].view(torch.int64))
                continue
            for key in ("status", "evals", "win_idx"):
                ok = ok and torch.equal(r[key], o[key])
            for key in ("x", "f", "win_x", "win_f", "win_key"):
                ok = ok and torch.equal(r[key].view(torch.int64), o[key].view(torch.int64))
        print(f"round {it:3d} {name:10s} {mode:7s} early={int(early)} jobs={len(jobs)} knobs={knobs} -> {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += 0 if ok else 1
    print("fuzz", "ok" if bad == 0 else f"FAILED ({bad} rounds)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

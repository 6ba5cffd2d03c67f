#!/usr/bin/env python3
"""Soak of the early-return path: many single ik() calls under the first-success rule back to back, interleaved with
batches, fk calls and deterministic calls on the same robot; every answer must reach its target (FK round trip).
Usage: python tools/first_success_soak.py [calls]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optik_amd import Robot, SolverConfig  # noqa: E402


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    r = Robot.from_urdf_file(os.path.join(ROOT, "optik_amd", "robots", "panda.urdf"), "panda_link0", "panda_link8")
    d = Robot.from_urdf_file(os.path.join(ROOT, "optik_amd", "robots", "panda.urdf"), "panda_link0", "panda_link8")
    d.set_parallelism(1)
    rng = np.random.default_rng(0)
    lb, ub = (np.array(v) for v in r.joint_limits())
    cfg = SolverConfig(max_time=0.0, max_restarts=4000)
    t0 = time.perf_counter()
    worst = 0.0
    for c in range(calls):
        tgt = np.array(r.fk(rng.uniform(lb, ub))) if c % 3 == 0 else tgt_keep
        tgt_keep = tgt
        x0 = rng.uniform(lb, ub)
        got = r.ik(cfg, tgt, x0.tolist())
        assert got is not None, c
        if c % 50 == 0:
            err = float(np.abs(np.array(r.fk(got[0])) - tgt).max())
            worst = max(worst, err)
            assert err < 5e-3, (c, err)
        if c % 500 == 7:
            out = r.ik_batch(SolverConfig(max_time=0.0, max_restarts=256), [tgt] * 8, np.tile(x0, (8, 1)))
            assert all(o is not None for o in out)
        if c % 700 == 11:
            assert d.ik(cfg, tgt, x0.tolist()) is not None
    dt = time.perf_counter() - t0
    print(f"{calls} first-success calls (+ batches, fk, deterministic calls in between) in {dt:.1f} s = {1e6 * dt / calls:.0f} us per "
          f"iteration of the loop; worst sampled FK error {worst:.2e}")


if __name__ == "__main__":
    main()

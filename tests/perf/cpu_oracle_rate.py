#!/usr/bin/env python3
"""CPU oracle rate on this host at 1 thread, half and all usable cores:
restarts/s and objective+gradient evaluations/s, Panda, the bench target.  Checker code timed
as a baseline only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    from bench import usable_cores
    from oracle import binding as ob, urdf_chain
    d = urdf_chain.chain_from_urdf(open(os.path.join(ROOT, "optik_amd", "robots", "panda.urdf")).read(),
                                   "panda_link0", "panda_link8")
    ch = ob.make_chain(**d)
    cfg = ob.make_config(solution_mode="speed", tol_f=1e-6)
    rng = np.random.default_rng(0)
    lb, ub = np.array(d["lb"]), np.array(d["ub"])
    target = ob.fk(ch, rng.uniform(lb, ub))[1]
    x0 = rng.uniform(lb, ub)
    cores = usable_cores()
    for th in sorted({1, max(1, cores // 2), cores}):
        n = 20000 * th
        t0 = time.perf_counter()
        res = ob.ik(ch, cfg, target, x0, 0, n, n_threads=th, early_exit=False, per_restart=True)
        dt = time.perf_counter() - t0
        print(f"{th:3d} threads: {n / dt:10,.0f} restarts/s  {res['evals'].sum() / dt:12,.0f} evaluations/s "
              f"({res['evals'].mean():.1f} per restart, {100.0 * res['success'].mean():.1f} % succeed)")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""CPU side of BASELINE config 5: T independent ik() calls (reachable random targets, random
seeds, SolutionMode::Speed, up to 256 restarts each, early exit) on the CPU oracle, one target
per call, on the host's usable cores.  Prints ik() calls/s.  The oracle is checker code; it is
only timed here as a baseline (the GPU figure: batch_ik_bench.py (a rounds 3-5 tool: git history))."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    from bench import usable_cores
    from oracle import binding as ob, urdf_chain
    d = urdf_chain.chain_from_urdf(open(os.path.join(ROOT, "optik_amd", "robots", "panda.urdf")).read(),
                                   "panda_link0", "panda_link8")
    ch = ob.make_chain(**d)
    rng = np.random.default_rng(0)
    lb, ub = np.array(d["lb"]), np.array(d["ub"])
    pose = [ob.fk(ch, rng.uniform(lb, ub))[1] for _ in range(T)]
    x0s = rng.uniform(lb, ub, size=(T, len(lb)))
    ocfg = ob.make_config(solution_mode="speed", max_restarts=R)
    cores = usable_cores()
    out = [None] * T

    def work(lo, hi):
        for t in range(lo, hi):
            out[t] = ob.ik(ch, ocfg, pose[t], x0s[t], 0, R, n_threads=1, early_exit=True)["found"]
    th = [threading.Thread(target=work, args=(k * T // cores, (k + 1) * T // cores)) for k in range(cores)]
    t0 = time.perf_counter()
    [x.start() for x in th]; [x.join() for x in th]
    dt = time.perf_counter() - t0
    print(f"CPU  oracle, {cores} threads (one target per call, early exit): {T} targets: {dt*1e3:.1f} ms -> "
          f"{T/dt:,.0f} ik() calls/s, {100.0*sum(bool(v) for v in out)/T:.1f} % solved")


if __name__ == "__main__":
    main()

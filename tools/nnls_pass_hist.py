#!/usr/bin/env python3
"""How the bounded sub-problems of the bench workload end (CPU oracle): solve passes x final size of the active set.
Usage: python tools/nnls_pass_hist.py [restarts]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob, urdf_chain  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = urdf_chain.chain_from_urdf(open(os.path.join(ROOT, "optik_amd", "robots", "panda.urdf")).read(), "panda_link0", "panda_link8")
ch = ob.make_chain(**d)
rng = np.random.default_rng(0)
lib = ob.lib()
h = (C.c_ulonglong * 256)()
lib.ok_nnls_hist(h)  # (clear)
evals = 0
for t in range(4):
    _, tgt = ob.fk(ch, rng.uniform(d["lb"], d["ub"]))
    x0 = rng.uniform(d["lb"], d["ub"])
    r = ob.ik(ch, ob.make_config(solution_mode="speed", tol_f=1e-6), tgt, x0, 0, R, n_threads=8, early_exit=False, per_restart=True)
    evals += int(r["evals"].sum())
lib.ok_nnls_hist(h)
a = np.array(list(h), dtype=np.float64).reshape(16, 16)
tot = a.sum()
print(f"{4 * R} restarts, {evals} evaluations, {int(tot)} bounded sub-problems ({tot / evals:.2f} per evaluation)")
print("by solve passes (1..15+), %:", " ".join(f"{100 * a[i].sum() / tot:.1f}" for i in range(1, 16)))
print("by final active-set size (0..8), %:", " ".join(f"{100 * a[:, j].sum() / tot:.1f}" for j in range(0, 9)))
print(f"one pass AND one active bound: {100 * a[1, 1] / tot:.1f} %; passes == active-set size (no removals): "
      f"{100 * sum(a[k, k] for k in range(1, 15)) / tot:.1f} %")
print(f"mean passes {sum(i * a[i].sum() for i in range(16)) / tot:.2f}")

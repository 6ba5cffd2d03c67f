#!/usr/bin/env python3
"""Generates tests/golden/generated/*.json from the CPU oracle (oracle/optik_oracle.c).

What is pinned and how:

* ``rng.json``      the ChaCha key of ``ChaCha8Rng::seed_from_u64(42)`` (lib.rs:360) and, for
                    each robot, the seeds of restarts 1..16 (``set_stream(i)`` + one
                    ``random_range(lb..=ub)`` per joint, lib.rs:86-91, 362, 369) under both
                    readings of rand 0.9.2's ``random_range`` (see optik_oracle.c:ok_uniform_scale).
* ``<robot>.inputs.txt``  the target pose and the caller seed of ``<robot>.json`` as plain text.
* ``<robot>.json``  one target, one caller seed x0, the default ``SolverConfig`` (tol_f = 1e-6,
                    tol_df = tol_dx = -1): for restarts 0..63 the seed, NLopt status, evaluation
                    count, returned x and f; the winners of ``SolutionMode::Speed`` (lowest
                    successful index = the 1-thread order) and ``SolutionMode::Quality``
                    (min ||x - x0||, lib.rs:397-413); the same again with tol_f = 1e-12
                    (tests/test_ik.rs:99).  Under both rules.

These are outputs of THIS repository's oracle, not of the reference (no Rust toolchain in
the build container): they make the oracle, the HIP path and -- through the Rust program in
INTEGRATION.md section 5, whose output ``tools/compare_golden.py`` reads -- the real optik
comparable on identical inputs.  Floats are written with ``repr`` (shortest round-trip), so the
files are bit-exact.

Usage: python tools/gen_golden.py          (rewrites the files; deterministic)
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import binding as ob  # noqa: E402
from oracle import urdf_chain  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "generated")
ROBOTS = {
    # name: (urdf path relative to the repo root, base link, ee link, harness seed)
    "ur3e": ("tests/golden/reference/ur3e.urdf", "ur_base_link", "ur_ee_link", 20),
    "panda": ("optik_amd/robots/panda.urdf", "panda_link0", "panda_link8", 18),
    "ur10": ("optik_amd/robots/ur10.urdf", "base_link", "ee_link", 29),
    # ten joints: two ChaCha blocks per restart seed, the general kernels of ik_wide.hpp on the GPU
    "arm10": ("tests/golden/robots/arm10.urdf", "l0", "l11", 31),
}
RULES = {"single_inclusive": ob.RANGE_SINGLE_INCLUSIVE, "new_inclusive": ob.RANGE_NEW_INCLUSIVE}
N_RESTARTS = 64
N_SEEDS = 16


def pose7_to_mat4_rowmajor(p):
    """[t, quat(i,j,k,w)] -> 4x4 nested list, row-major (what optik-py's parse_pose takes)."""
    t, (i, j, k, w) = p[:3], p[3:]
    R = [[w * w + i * i - j * j - k * k, 2 * (i * j - w * k), 2 * (w * j + i * k)],
         [2 * (w * k + i * j), w * w - i * i + j * j - k * k, 2 * (j * k - w * i)],
         [2 * (i * k - w * j), 2 * (w * i + j * k), w * w - i * i - j * j + k * k]]
    return [R[0] + [t[0]], R[1] + [t[1]], R[2] + [t[2]], [0.0, 0.0, 0.0, 1.0]]


def load(name):
    path, base, ee, seed = ROBOTS[name]
    with open(os.path.join(ROOT, path)) as fh:
        d = urdf_chain.chain_from_urdf(fh.read(), base, ee)
    return d, ob.make_chain(**d), seed


def solve_block(ch, cfgs, tgt, x0):
    per = ob.ik(ch, cfgs["speed"], tgt, x0, 0, N_RESTARTS, n_threads=1, early_exit=False, per_restart=True)
    restarts = []
    for i in range(N_RESTARTS):
        seed = x0 if i == 0 else ob.restart_seed(ch, i)
        restarts.append(dict(index=i, seed=[float(v) for v in seed], status=int(per["status"][i]),
                             success=bool(per["success"][i]), evals=int(per["evals"][i]),
                             x=[float(v) for v in per["xs"][i]], f=float(per["fs"][i])))
    winners = {}
    for mode in ("speed", "quality"):
        w = ob.ik(ch, cfgs[mode], tgt, x0, 0, N_RESTARTS, n_threads=1, early_exit=False)
        winners[mode] = dict(found=bool(w["found"]), index=int(w["winner"]) if w["found"] else None,
                             x=[float(v) for v in w["x"]] if w["found"] else None,
                             f=float(w["f"]) if w["found"] else None)
    return dict(restarts=restarts, winners=winners)


def main():
    os.makedirs(OUT, exist_ok=True)
    ob.build()
    import ctypes as C
    key = (C.c_uint32 * 8)()
    ob.lib().ok_seed_from_u64(42, key)
    rng_doc = dict(
        what="ChaCha8Rng::seed_from_u64(42) key bytes (hex) and restart seeds 1..16 per robot and rule",
        generated_by="tools/gen_golden.py (CPU oracle of this repository; NOT an output of the reference)",
        seed_from_u64_42_key_hex=bytes(key).hex(), robots={})
    for name in ROBOTS:
        d, ch, hseed = load(name)
        rng = np.random.default_rng(hseed)
        q_star = rng.uniform(d["lb"], d["ub"])
        x0 = rng.uniform(d["lb"], d["ub"])
        _, tgt = ob.fk(ch, q_star)
        doc = dict(
            generated_by="tools/gen_golden.py (CPU oracle of this repository; NOT an output of the reference)",
            robot=name, urdf=ROBOTS[name][0], base_link=ROBOTS[name][1], ee_link=ROBOTS[name][2],
            lb=[float(v) for v in d["lb"]], ub=[float(v) for v in d["ub"]],
            q_target=[float(v) for v in q_star], target_pose7=[float(v) for v in tgt],
            target_mat4_rowmajor=pose7_to_mat4_rowmajor([float(v) for v in tgt]),
            x0=[float(v) for v in x0], n_restarts=N_RESTARTS, rules={})
        rng_doc["robots"][name] = {}
        for rname, rule in RULES.items():
            with ob.range_rule(rule):
                rng_doc["robots"][name][rname] = [
                    [float(v) for v in ob.restart_seed(ch, i)] for i in range(1, N_SEEDS + 1)]
                blocks = {}
                for label, tol_f in (("tol_f_1e-6", 1e-6), ("tol_f_1e-12", 1e-12)):
                    cfgs = {m: ob.make_config(solution_mode=m, tol_f=tol_f) for m in ("speed", "quality")}
                    blocks[label] = solve_block(ch, cfgs, tgt, x0)
                doc["rules"][rname] = blocks
        with open(os.path.join(OUT, f"{name}.json"), "w") as fh:
            json.dump(doc, fh, indent=0, separators=(",", ":"))
            fh.write("\n")
        # the inputs alone, as plain decimal text (exact: repr round-trips), for the Rust
        # program of INTEGRATION.md section 5: line 1 = target [tx ty tz qi qj qk qw], line 2 = x0
        with open(os.path.join(OUT, f"{name}.inputs.txt"), "w") as fh:
            fh.write(" ".join(repr(float(v)) for v in tgt) + "\n")
            fh.write(" ".join(repr(float(v)) for v in x0) + "\n")
    with open(os.path.join(OUT, "rng.json"), "w") as fh:
        json.dump(rng_doc, fh, indent=0, separators=(",", ":"))
        fh.write("\n")
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()

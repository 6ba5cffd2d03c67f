#!/usr/bin/env python3
"""How much does the platform's libm matter?  (CPU only; the oracle is the instrument, nothing here is product code.)

The reference calls the platform's `f64::sin_cos` / `f64::atan2` (/root/reference/crates/optik/src/math.rs:54,76,113,144;
nalgebra's `from_axis_angle` under kinematics.rs:245-248).  The oracle -- and with it the GPU kernels -- use their own
fdlibm operation sequence instead, so that "GPU == oracle" can be bit-exact.  Both are < 1 ulp
accurate, so they differ from glibc in the last bit of some calls; random-restart SLSQP is chaotic in roundoff, so a
last-bit difference can end a restart somewhere else.  This tool runs the SAME restarts through the oracle twice -- its
own sin / cos / atan2, and glibc's (`-DOK_PLATFORM_LIBM`: what a Linux build of the reference binds) -- and reports how
often the outcome differs: the error bar on "results match the reference" that does not depend on any restatement
being wrong.

    python tools/libm_sensitivity.py [--scale 1.0] [--json out.json]

--scale < 1 shrinks every workload (tests/test_oracle_libm_sensitivity.py runs 1/64).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

ROBOTS = {"panda": ("panda.urdf", "panda_link0", "panda_link8"), "ur10": ("ur10.urdf", "base_link", "ee_link")}


def _threads():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


def _chain(ob, name):
    from oracle import urdf_chain
    f, base, ee = ROBOTS[name]
    d = urdf_chain.chain_from_urdf(open(os.path.join(ROOT, "optik_amd", "robots", f)).read(), base, ee)
    return ob.make_chain(**d), np.array(d["lb"]), np.array(d["ub"])


def _inputs(ob, name, T, seed):
    """Reachable targets FK(q*) and seeds, q* and x0 uniform in the limits (the full-size GPU tests' recipe; the targets
    are formed ONCE, with the oracle's default build, and the same seven numbers go to both runs)."""
    ob.use_portable_build()
    ch, lb, ub = _chain(ob, name)
    rng = np.random.default_rng(seed)
    q = rng.uniform(lb, ub, size=(T, len(lb)))
    tg = np.array([ob.fk(ch, q[t])[1] for t in range(T)])
    x0 = rng.uniform(lb, ub, size=(T, len(lb)))
    return tg, x0


def _run(ob, which, name, cfg_kw, tg, x0, R):
    """Every restart 0..R-1 of every target, to termination, under one libm."""
    (ob.use_libm_build if which == "glibc" else ob.use_native_build)()
    ch, _, _ = _chain(ob, name)
    cfg = ob.make_config(**cfg_kw)
    out = []
    for t in range(len(tg)):
        out.append(ob.ik(ch, cfg, tg[t], x0[t], 0, R, n_threads=_threads(), early_exit=False, per_restart=True))
    return out


def _winners(res, x0):
    """Speed: lowest solved index; Quality: solved restart closest to the seed (lib.rs:397-413), -1: none."""
    ok = res["success"] != 0
    if not ok.any():
        return -1, -1
    idx = np.flatnonzero(ok)
    d = np.linalg.norm(res["xs"][idx] - x0[None, :], axis=1)
    return int(idx[0]), int(idx[np.argmin(d)])


def compare(a, b, x0):
    """Per-restart and per-target differences between two runs of the same restarts."""
    n = sum(len(r["status"]) for r in a)
    st = sum(int((ra["status"] != rb["status"]).sum()) for ra, rb in zip(a, b))
    succ = sum(int((ra["success"] != rb["success"]).sum()) for ra, rb in zip(a, b))
    ev = sum(int((ra["evals"] != rb["evals"]).sum()) for ra, rb in zip(a, b))
    dx = np.concatenate([np.abs(ra["xs"] - rb["xs"]).max(axis=1) for ra, rb in zip(a, b)])
    bits = sum(int(((ra["xs"].view(np.int64) != rb["xs"].view(np.int64)).any(axis=1)).sum()) for ra, rb in zip(a, b))
    both_ok = np.concatenate([(ra["success"] != 0) & (rb["success"] != 0) for ra, rb in zip(a, b)])
    wa = [_winners(r, x0[t]) for t, r in enumerate(a)]
    wb = [_winners(r, x0[t]) for t, r in enumerate(b)]
    T = len(a)
    # a changed winner is still a solution: how far apart are the two answers
    qdx = []
    for t in range(T):
        if wa[t][1] >= 0 and wb[t][1] >= 0:
            qdx.append(float(np.abs(a[t]["xs"][wa[t][1]] - b[t]["xs"][wb[t][1]]).max()))
    return {
        "restarts": n, "targets": T,
        "frac_status_differs": st / n, "frac_success_differs": succ / n, "frac_evals_differ": ev / n,
        "frac_x_not_bit_equal": bits / n,
        "frac_x_differs_gt_1e-6": float((dx > 1e-6).mean()),
        "frac_x_differs_gt_1e-6_among_both_solved": float((dx[both_ok] > 1e-6).mean()) if both_ok.any() else None,
        "median_abs_dx_both_solved": float(np.median(dx[both_ok])) if both_ok.any() else None,
        "speed_winner_index_changes": sum(x[0] != y[0] for x, y in zip(wa, wb)) / T,
        "quality_winner_index_changes": sum(x[1] != y[1] for x, y in zip(wa, wb)) / T,
        "quality_winner_x_differs_gt_1e-6": (float(np.mean(np.array(qdx) > 1e-6)) if qdx else None),
        "solve_rate": [float(np.mean([w[0] >= 0 for w in wa])), float(np.mean([w[0] >= 0 for w in wb]))],
    }


def study(scale=1.0, verbose=True):
    from oracle import binding as ob
    cases = [
        # BASELINE config 2: Panda, one target, 65 536 restarts, tol_f 1e-6
        ("config2_panda_65536", "panda", dict(solution_mode="speed", tol_f=1e-6), 1, 65536, 0),
        # BASELINE config 3: UR10, one target, 2^20 restarts, tol_f 1e-12 (tests/test_ik.rs:99's tolerance)
        ("config3_ur10_1M_tol1e-12", "ur10", dict(solution_mode="quality", tol_f=1e-12), 1, 1 << 20, 3),
        # one GPU's share of BASELINE config 5: 512 Panda targets x 256 restarts -- 512 winners of each mode
        ("config5_share_512x256", "panda", dict(solution_mode="speed", tol_f=1e-6), 512, 256, 5),
    ]
    out = {"threads": _threads(), "scale": scale, "cases": {}}
    try:
        for name, robot, cfg_kw, T, R, seed in cases:
            if T == 1:
                R = max(256, int(R * scale))
            else:
                T = max(8, int(T * scale))
            t0 = time.perf_counter()
            tg, x0 = _inputs(ob, robot, T, seed)
            a = _run(ob, "oracle", robot, cfg_kw, tg, x0, R)
            b = _run(ob, "glibc", robot, cfg_kw, tg, x0, R)
            rec = compare(a, b, x0)
            rec["seconds"] = time.perf_counter() - t0
            rec["workload"] = f"{robot}, {T} target(s) x restarts 0..{R - 1}, tol_f {cfg_kw['tol_f']:g}, every restart to termination"
            out["cases"][name] = rec
            if verbose:
                print(name, json.dumps(rec), flush=True)
    finally:
        ob.use_portable_build()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    out = study(args.scale)
    if args.json:
        with open(args.json, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Fuzz of the general solver (optik_amd/csrc/ik_wide.hpp): random 9 .. 16-joint arms (and, through
option solve_kernel = general, random sub-chains of at most 8 joints), random targets / seeds / tolerances /
weights / ee offsets / restart ranges / launch sizes (both forms of the solver), Speed with and without early
exit -- every restart's status, evaluation count, x and f against the CPU oracle, bit for bit, and the winners.
Usage: python tools/wide_fuzz.py [rounds] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_wide_robots  # noqa: E402
from optik_amd import _native as nat  # noqa: E402
from optik_amd import device  # noqa: E402
from oracle import binding as ob  # noqa: E402
from oracle import urdf_chain  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(((bits(a) == bits(b)) | ((a == 0) & (b == 0)) | (np.isnan(a) & np.isnan(b))).all())


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ob.build()
    for it in range(rounds):
        small = rng.random() < 0.25
        n = int(rng.integers(3, 9)) if small else int(rng.integers(9, 17))
        tip = bool(rng.integers(0, 2))
        d = urdf_chain.chain_from_urdf(gen_wide_robots.arm(n, tip, int(rng.integers(0, 1 << 30))), "l0",
                                       f"l{n + (1 if tip else 0)}")
        ch = ob.make_chain(**d)
        hc = device.HipChain(**d)
        T = int(rng.choice([1, 1, 2, 5]))
        R = int(rng.choice([37, 300, 1100, 2500, 6000])) // T + 1
        begin = int(rng.choice([0, 0, 1, 1000, 2**33 + 5]))
        tg, x0 = [], []
        for _ in range(T):
            _, ee = ob.fk(ch, rng.uniform(d["lb"], d["ub"]))
            tg.append(ee)
            x0.append(rng.uniform(d["lb"], d["ub"]))
        tg, x0 = np.array(tg), np.array(x0)
        kw = dict(solution_mode=str(rng.choice(["speed", "quality"])), tol_f=10.0 ** -int(rng.integers(4, 12)))
        if rng.random() < 0.3:
            kw.update(tol_df=10.0 ** -int(rng.integers(8, 16)), tol_dx=10.0 ** -int(rng.integers(8, 14)))
        if rng.random() < 0.4:
            kw.update(linear_weight=tuple(rng.uniform(0.1, 3, 3)), angular_weight=tuple(rng.uniform(0.1, 3, 3)))
        ee_off = None
        if rng.random() < 0.4:
            q = rng.normal(size=4)
            ee_off = np.concatenate([rng.uniform(-0.1, 0.1, 3), q / np.linalg.norm(q)])
        early = kw["solution_mode"] == "speed" and rng.random() < 0.5
        forced = str(rng.choice(["", "", "lds", "hbm"]))  # (the scheduler's own choice, or one of the two forms)
        opts = {}
        if small:
            opts["solve_kernel"] = "general"
        if forced:
            opts["wide_form"] = forced
        with nat.options(**opts):
            out = hc.ik_batch(nat.make_config(**kw), torch.tensor(tg, device="cuda"), torch.tensor(x0, device="cuda"),
                              begin, begin + R, flags=nat.IK_EARLY_EXIT if early else 0, ee_offset7=ee_off)
            torch.cuda.synchronize()
        form = "lds" if hc.last_launch()["lds_bytes"] > 8192 else "hbm"
        ok = True
        st = out["status"].cpu().numpy().reshape(T, R)
        ev = out["evals"].cpu().numpy().reshape(T, R)
        fs = out["f"].cpu().numpy().reshape(T, R)
        xs = out["x"].cpu().numpy()
        for t in range(T):
            ref = ob.ik(ch, ob.make_config(**kw), tg[t], x0[t], begin, begin + R, n_threads=4, early_exit=False,
                        per_restart=True, ee_offset=ob.Pose.make(ee_off[:3], ee_off[3:]) if ee_off is not None else None)
            win = int(out["win_idx"].cpu()[t])
            ok = ok and win == (ref["winner"] if ref["found"] else -1)
            if ref["found"]:
                ok = ok and same(out["win_x"].cpu().numpy()[t], ref["x"])
            if early:
                # abandoned restarts (index above the winner) carry FORCED_STOP; the others equal the oracle's
                keep = st[t] != nat.RES_FORCED_STOP
                ok = ok and (not ref["found"] or bool(keep[: ref["winner"] - begin + 1].all()))
            else:
                keep = np.ones(R, dtype=bool)
            ok = ok and np.array_equal(st[t][keep], ref["status"][keep]) and np.array_equal(ev[t][keep], ref["evals"][keep])
            ok = ok and same(fs[t][keep], ref["fs"][keep]) and same(xs[:, t * R:(t + 1) * R][:, keep], ref["xs"].T[:, keep])
        print(f"round {it} n={n} tip={int(tip)} T={T} R={R} begin={begin} {kw['solution_mode']} early={int(early)} "
              f"form={form}{' (general solver, n <= 8)' if small else ''} -> {'ok' if ok else 'MISMATCH'}", flush=True)
        if not ok:
            raise SystemExit(1)
    print("fuzz ok")


if __name__ == "__main__":
    main()

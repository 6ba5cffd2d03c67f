#!/usr/bin/env python3
"""Fuzz of the tuned single-launch solvers (ik_lane64.hpp, ik_quad.hpp) through optik_hip_ik_batch: random robots
(UR3e, Panda, Panda + hand, UR10, Panda sub-chains, the 8-joint arm), targets / seeds / tolerances / weights /
ee offsets / restart ranges / target counts / hand-out order, Speed with and without early exit (both readings of
should_exit), on the solver a launch of that size gets, the quad solver forced and the lane-per-restart form forced
-- every restart's status, evaluation count, x and f against the CPU oracle, bit for bit, and the winners.
(Rounds 1-4 had this as tools/engine_fuzz.py against the streaming engine's scheduling knobs.)
Usage: python tools/solver_fuzz.py [rounds] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ROBOT_SPECS  # noqa: E402
from optik_amd import _native as nat  # noqa: E402
from optik_amd import device  # noqa: E402
from oracle import binding as ob  # noqa: E402
from oracle import urdf_chain  # noqa: E402

NAMES = ["ur3e", "panda", "panda_hand", "ur10", "panda5", "panda3", "arm8"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(((bits(a) == bits(b)) | ((a == 0) & (b == 0)) | (np.isnan(a) & np.isnan(b))).all())


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ob.build()
    robots = {}
    for name in NAMES:
        path, base, ee = ROBOT_SPECS[name]
        with open(path) as fh:
            d = urdf_chain.chain_from_urdf(fh.read(), base, ee)
        robots[name] = (d, ob.make_chain(**d), device.HipChain(**d))
    for it in range(rounds):
        name = str(rng.choice(NAMES))
        d, ch, hc = robots[name]
        n = len(d["lb"])
        T = int(rng.choice([1, 1, 1, 3, 17]))
        R = int(rng.choice([50, 700, 3000, 9000, 70000])) // T + 1
        begin = int(rng.choice([0, 0, 1, 4097, 2**33 + 5]))
        tg, x0 = [], []
        for _ in range(T):
            _, ee = ob.fk(ch, rng.uniform(d["lb"], d["ub"]))
            tg.append(ee)
            x0.append(rng.uniform(d["lb"], d["ub"]))
        tg, x0 = np.array(tg), np.array(x0)
        kw = dict(solution_mode=str(rng.choice(["speed", "quality"])), tol_f=10.0 ** -int(rng.integers(4, 13)))
        if rng.random() < 0.3:
            kw.update(tol_df=10.0 ** -int(rng.integers(8, 16)), tol_dx=10.0 ** -int(rng.integers(8, 14)))
        if rng.random() < 0.4:
            kw.update(linear_weight=tuple(rng.uniform(0.1, 3, 3)), angular_weight=tuple(rng.uniform(0.1, 3, 3)))
        ee_off = None
        if rng.random() < 0.4:
            q = rng.normal(size=4)
            ee_off = np.concatenate([rng.uniform(-0.1, 0.1, 3), q / np.linalg.norm(q)])
        early = kw["solution_mode"] == "speed" and rng.random() < 0.5
        find_any = early and rng.random() < 0.3
        flags = (nat.IK_EARLY_EXIT if early else 0) | (nat.IK_FIND_ANY if find_any else 0)
        if T > 1 and rng.random() < 0.5:
            flags |= nat.IK_RESTART_MAJOR
        solver = str(rng.choice(["auto", "quad", "lane64"]))
        with nat.options(**({} if solver == "auto" else {"solve_kernel": solver})):
            out = hc.ik_batch(nat.make_config(**kw), torch.tensor(tg, device="cuda"), torch.tensor(x0, device="cuda"),
                              begin, begin + R, flags=flags, ee_offset7=ee_off)
            torch.cuda.synchronize()
        ran = "lane64" if hc.last_launch()["lds_bytes"] > 30000 else "quad"
        ok = True
        st = out["status"].cpu().numpy().reshape(T, R)
        ev = out["evals"].cpu().numpy().reshape(T, R)
        fs = out["f"].cpu().numpy().reshape(T, R)
        xs = out["x"].cpu().numpy()
        for t in range(T):
            ref = ob.ik(ch, ob.make_config(**kw), tg[t], x0[t], begin, begin + R, n_threads=8, early_exit=False,
                        per_restart=True, ee_offset=ob.Pose.make(ee_off[:3], ee_off[3:]) if ee_off is not None else None)
            win = int(out["win_idx"].cpu()[t])
            if find_any:
                # any success may win: it must BE a success of the oracle's, with that restart's numbers
                ok = ok and ((win >= 0) == bool(ref["found"]))
                if win >= 0:
                    ok = ok and ref["success"][win - begin] != 0 and same(out["win_x"].cpu().numpy()[t], ref["xs"][win - begin])
            else:
                ok = ok and win == (ref["winner"] if ref["found"] else -1)
                if ref["found"]:
                    ok = ok and same(out["win_x"].cpu().numpy()[t], ref["x"])
            if early:
                # abandoned restarts carry FORCED_STOP; the others equal the oracle's; nothing below the deterministic
                # winner is abandoned under the deterministic rule
                keep = st[t] != nat.RES_FORCED_STOP
                if not find_any:
                    ok = ok and (not ref["found"] or bool(keep[: ref["winner"] - begin + 1].all()))
            else:
                keep = np.ones(R, dtype=bool)
            ok = ok and np.array_equal(st[t][keep], ref["status"][keep]) and np.array_equal(ev[t][keep], ref["evals"][keep])
            ok = ok and same(fs[t][keep], ref["fs"][keep]) and same(xs[:, t * R:(t + 1) * R][:, keep], ref["xs"].T[:, keep])
        print(f"round {it} {name} T={T} R={R} begin={begin} {kw['solution_mode']} early={int(early)} any={int(find_any)} "
              f"flags={flags} asked={solver} ran={ran} -> {'ok' if ok else 'MISMATCH'}", flush=True)
        if not ok:
            raise SystemExit(1)
    print("fuzz ok")


if __name__ == "__main__":
    main()

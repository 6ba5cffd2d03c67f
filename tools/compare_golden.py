#!/usr/bin/env python3
"""Compares the output of the Rust program of INTEGRATION.md section 5 (real optik, run by
someone with cargo) with tests/golden/generated/<robot>.json + rng.json.

    python tools/compare_golden.py panda panda.rust.txt

Reports, per item, whether the real crate agrees with the committed fixture under the
``single_inclusive`` rule, the ``new_inclusive`` rule, both or neither: bit-exact for the key
and the seeds; for solve results bit-exact and within the north-star tolerance (1e-6 on joint
angles) separately.  Exit code 0 iff one rule agrees everywhere within 1e-6 and the Speed /
Quality winners are the fixture's winners.  This is the step that turns DESIGN.md's "parity
unpinned" (SLSQP + RNG restated from un-vendored crates) into "pinned".
"""
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "tests", "golden", "generated")


def parse(path):
    out = dict(key=None, seeds={}, restarts={}, speed={}, quality={})
    num = r"[-+0-9.eE]+|inf|NaN"
    for line in open(path):
        line = line.strip()
        if not line:
            continue
        tag, rest = line.split(" ", 1)
        if tag == "key":
            out["key"] = "".join(re.findall(r"[0-9a-fA-F]{2}", rest)).lower()
            continue
        if tag == "seed":
            i, vec = rest.split(" ", 1)
            out["seeds"][int(i)] = [float(v) for v in re.findall(num, vec)]
            continue
        parts = rest.split(" ", 2 if tag == "restart" else 1)
        tol = float(parts[0])
        body = parts[-1]
        if tag == "restart":
            idx = int(parts[1])
        if body.strip() == "none":
            val = None
        else:
            vals = [float(v) for v in re.findall(num, body)]
            val = (vals[:-1], vals[-1])
        if tag == "restart":
            out["restarts"][(tol, idx)] = val
        else:
            out[tag][tol] = val
    return out


def main():
    robot, path = sys.argv[1], sys.argv[2]
    got = parse(path)
    doc = json.load(open(os.path.join(GEN, f"{robot}.json")))
    rng = json.load(open(os.path.join(GEN, "rng.json")))
    ok_all = {}
    print("key:", "MATCH" if got["key"] == rng["seed_from_u64_42_key_hex"] else
          f"DIFFERENT (crate {got['key']}, fixture {rng['seed_from_u64_42_key_hex']})")
    for rule in ("single_inclusive", "new_inclusive"):
        want = rng["robots"][robot][rule]
        exact = sum(np.array_equal(np.array(got["seeds"][i + 1]).view(np.uint64), np.array(w).view(np.uint64))
                    for i, w in enumerate(want) if i + 1 in got["seeds"])
        print(f"seeds under {rule}: {exact}/{len(want)} bit-exact")
        worst, n_exact, n_close, n_cls, n = 0.0, 0, 0, 0, 0
        winners_ok = True
        for label, blk in doc["rules"][rule].items():
            tol = float(label.split("_")[-1])
            for r in blk["restarts"]:
                g = got["restarts"].get((tol, r["index"]), "missing")
                if g == "missing":
                    continue
                n += 1
                if (g is not None) != r["success"]:
                    continue
                n_cls += 1
                if g is None:
                    n_exact += 1
                    n_close += 1
                    continue
                d = float(np.max(np.abs(np.array(g[0]) - np.array(r["x"]))))
                worst = max(worst, d)
                n_close += d <= 1e-6
                n_exact += np.array_equal(np.array(g[0]).view(np.uint64), np.array(r["x"]).view(np.uint64))
            for mode in ("speed", "quality"):
                g, w = got[mode].get(tol), blk["winners"][mode]
                same = (g is None and not w["found"]) or (g is not None and w["found"]
                                                         and np.max(np.abs(np.array(g[0]) - np.array(w["x"]))) <= 1e-6)
                winners_ok = winners_ok and bool(same)
                print(f"  {label} {mode} winner (fixture index {w['index']}):", "same" if same else "DIFFERENT")
        print(f"restarts under {rule}: {n_cls}/{n} same success/failure, {n_close}/{n} within 1e-6, "
              f"{n_exact}/{n} bit-exact, worst |dx| {worst:.3e}")
        ok_all[rule] = (n > 0 and n_close == n and winners_ok)
    print("verdict:", {k: ("agrees" if v else "differs") for k, v in ok_all.items()})
    sys.exit(0 if any(ok_all.values()) else 1)


if __name__ == "__main__":
    main()

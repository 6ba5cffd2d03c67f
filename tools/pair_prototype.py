#!/usr/bin/env python3
"""Two lanes per restart, priced from compiled code.

tools/proto_pair/pair_proto.hip restates the two heaviest per-lane phases of the lane-per-restart solver -- the
objective + gradient evaluation and the BFGS update of the packed factor -- for a PAIR of lanes per restart, next to the
product's one-lane functions in the same object.  This script

  (no GPU)  compiles it for gfx950 and prints, per kernel: registers, scratch, and the instructions of the kernel body by
            class -- per wave and per RESTART (a pair wave holds 32 restarts, a lane wave 64);
  --run     (on the GPU box) checks that both forms give the same bits on random inputs and times them at the occupancy
            each form would have inside a solver: the one-lane kernels at ONE wave per SIMD (39 KB of LDS per single-wave
            workgroup, like ik_lane_kernel), the pair kernels at TWO (19 KB).

Usage: python tools/pair_prototype.py [--run] [--json out.json]
"""
import ctypes as C
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from optik_amd import build as pb  # noqa: E402
import inst_mix  # noqa: E402

SRC = os.path.join(ROOT, "tools", "proto_pair", "pair_proto.hip")
OUT = os.path.join(ROOT, "tools", "proto_pair", "libpair_proto.so")
KERNELS = {"eval_pair_kernel<7, true>": ("evaluation", "pair", 32), "eval_lane_kernel<7, true>": ("evaluation", "lane", 64),
           "bfgs_pair_kernel<7>": ("BFGS update", "pair", 32), "bfgs_lane_kernel<7>": ("BFGS update", "lane", 64)}


def compile_all():
    flags = [*pb.FLAGS, "-I", pb.CSRC]
    rem = subprocess.run([pb._hipcc(), *flags, "-Rpass-analysis=kernel-resource-usage", "-x", "hip", "--cuda-device-only", "-S", SRC,
                          "-o", "/tmp/pair_proto.s"], capture_output=True, text=True)
    if rem.returncode:
        raise SystemExit(rem.stderr[-3000:])
    res = pb.parse_resource_remarks(rem.stderr)
    text = open("/tmp/pair_proto.s").read()
    out = {}
    for name, (phase, form, per_wave) in KERNELS.items():
        key = next(k for k in res if k.replace("optik::", "") == name)
        mangled = None
        for m in re.finditer(r"^(_ZN5optik\w+):", text, re.M):
            dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout
            if dem.replace("void ", "").replace("optik::", "").startswith(name + "("):
                mangled = m.group(1)
        body = text[text.index("\n" + mangled + ":"):]
        body = body[:body.index(".Lfunc_end")]
        cnt = {}
        for ln in body.splitlines():
            m = re.match(r"^\s+([a-z][a-z0-9_]+)\s*(.*)$", ln)
            if m:
                c = inst_mix.classify(m.group(1), m.group(2))
                if c:
                    cnt[c] = cnt.get(c, 0) + 1
        valu = sum(cnt.get(k, 0) for k in ("f64", "cmp64", "select", "mov", "dpp", "agpr", "int"))
        out[name] = {"phase": phase, "form": form, "restarts_per_wave": per_wave, **res[key], "valu": valu, "f64": cnt.get("f64", 0),
                     "dpp": cnt.get("dpp", 0), "select": cnt.get("select", 0), "all_instructions": sum(cnt.values()),
                     "valu_per_restart": valu / per_wave, "all_per_restart": sum(cnt.values()) / per_wave}
    return out


def build_so():
    subprocess.check_call([pb._hipcc(), *pb.FLAGS, "-I", pb.CSRC, "-shared", "-x", "hip", SRC, "-o", OUT])
    return C.CDLL(OUT)


class ChainDev(C.Structure):
    _fields_ = [("n_pos", C.c_int32), ("has_tip", C.c_int32), ("pad0", C.c_int32), ("pad1", C.c_int32),
                ("origin", (C.c_double * 7) * 9), ("axis", (C.c_double * 3) * 8), ("lb", C.c_double * 8), ("ub", C.c_double * 8)]


def run(table):
    import numpy as np
    from oracle import urdf_chain
    L = build_so()
    dp = C.POINTER(C.c_double)
    d = urdf_chain.chain_from_urdf(open(os.path.join(ROOT, "optik_amd", "robots", "panda.urdf")).read(), "panda_link0", "panda_link8")
    ch = ChainDev()
    n = len(d["lb"])
    ch.n_pos, ch.has_tip = n, int(len(d["origins"]) > n)
    for j, o in enumerate(d["origins"]):
        ch.origin[j][:] = list(o)
    k = 0
    for j, t in enumerate(d["types"]):
        if t != 0 and k < n:   # positional joints carry the axes, in order
            ch.axis[k][:] = list(d["axes"][j])
            k += 1
    ch.lb[:n] = list(d["lb"]); ch.ub[:n] = list(d["ub"])
    rng = np.random.default_rng(0)
    B = 1 << 21
    q = np.ascontiguousarray(rng.uniform(np.array(d["lb"])[:, None], np.array(d["ub"])[:, None], size=(n, B)))
    target = np.array([0.4, 0.1, 0.5, 0.0, 1.0, 0.0, 0.0])
    res = {}
    outs = {}
    for pair in (0, 1):
        f, g = np.zeros(B), np.zeros((n, B))
        ms = C.c_float(0)
        rc = L.proto_eval(pair, ch.has_tip, C.byref(ch), target.ctypes.data_as(dp), q.ctypes.data_as(dp), C.c_longlong(B),
                          f.ctypes.data_as(dp), g.ctypes.data_as(dp), 10, C.byref(ms))
        assert rc == 0, rc
        outs[pair] = (f, g)
        res["eval_" + ("pair" if pair else "lane")] = {"ms": ms.value, "ns_per_evaluation": ms.value * 1e6 / B}
    same_f = np.array_equal(outs[0][0].view(np.int64), outs[1][0].view(np.int64))
    same_g = np.array_equal(outs[0][1].view(np.int64), outs[1][1].view(np.int64))
    res["eval_bits_equal"] = bool(same_f and same_g)
    # BFGS: a positive definite packed factor (unit lower L, positive D), a step and a gradient difference
    NL = n * (n + 1) // 2
    l = rng.uniform(-0.5, 0.5, size=(NL, B))
    idx = 0
    for c in range(n):
        l[idx] = rng.uniform(0.5, 2.0, size=B)   # the diagonal entry (column c starts at lidx(c, c))
        idx += n - c
    s = rng.uniform(-0.3, 0.3, size=(n, B))
    u = s * rng.uniform(0.2, 3.0, size=(n, B)) + rng.uniform(-0.05, 0.05, size=(n, B))
    l, s, u = (np.ascontiguousarray(a) for a in (l, s, u))
    louts = {}
    for pair in (0, 1):
        lo = np.zeros((NL, B))
        ms = C.c_float(0)
        rc = L.proto_bfgs(pair, l.ctypes.data_as(dp), s.ctypes.data_as(dp), u.ctypes.data_as(dp), C.c_longlong(B),
                          lo.ctypes.data_as(dp), 10, C.byref(ms))
        assert rc == 0, rc
        louts[pair] = lo
        res["bfgs_" + ("pair" if pair else "lane")] = {"ms": ms.value, "ns_per_update": ms.value * 1e6 / B}
    res["bfgs_bits_equal"] = bool(np.array_equal(louts[0].view(np.int64), louts[1].view(np.int64)))
    res["bfgs_nan_frac"] = float(np.isnan(louts[0]).mean())
    res["eval_time_ratio_pair_over_lane"] = res["eval_pair"]["ms"] / res["eval_lane"]["ms"]
    res["bfgs_time_ratio_pair_over_lane"] = res["bfgs_pair"]["ms"] / res["bfgs_lane"]["ms"]
    return res


def main():
    table = compile_all()
    print("%-28s %-12s %5s %5s %5s %8s %7s | %6s %6s %6s %6s | %9s %9s" % ("kernel", "phase", "form", "VGPR", "AGPR", "scratch", "waves", "VALU",
                                                                            "f64", "DPP", "select", "VALU/rst", "all/rst"))
    for name, r in table.items():
        print("%-28s %-12s %5s %5d %5d %8d %7d | %6d %6d %6d %6d | %9.1f %9.1f" % (name, r["phase"], r["form"], r["vgpr"], r["agpr"], r["scratch"],
                                                                                   r["occupancy"], r["valu"], r["f64"], r["dpp"], r["select"],
                                                                                   r["valu_per_restart"], r["all_per_restart"]))
    for phase in ("evaluation", "BFGS update"):
        a = next(r for r in table.values() if r["phase"] == phase and r["form"] == "pair")
        b = next(r for r in table.values() if r["phase"] == phase and r["form"] == "lane")
        print(f"{phase}: VALU instructions per restart, pair / lane = {a['valu_per_restart'] / b['valu_per_restart']:.2f}; "
              f"all instructions {a['all_per_restart'] / b['all_per_restart']:.2f}")
    doc = {"static": table}
    if "--run" in sys.argv:
        doc["measured"] = run(table)
        print(json.dumps(doc["measured"], indent=1))
    if "--json" in sys.argv:
        json.dump(doc, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()

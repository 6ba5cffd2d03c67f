#!/usr/bin/env python3
"""Static instruction mix of ik_lane_kernel<7, true> by phase of a trip (no GPU needed).

Compiles the lane kernel's translation unit to ISA with the product's flags plus -gline-tables-only (line tables do
not change the generated code: the instruction count is checked against the plain build), attributes every instruction
to the source region its .loc points at, and classes it:

    f64      v_add / v_mul / v_fma(c) / v_div_* / v_rcp / v_rsq / v_sqrt / v_ldexp / v_frexp / v_min / v_max / rounding, _f64
    cmp64    v_cmp_*_f64, v_cmp_class_f64
    select   v_cndmask_b32
    mov      v_mov_b32 / v_mov_b64 (no DPP)
    dpp      any VALU instruction with a DPP control (quad_perm / row_*), v_permlane*, ds_bpermute / ds_swizzle
    agpr     v_accvgpr_read / v_accvgpr_write (the register allocator's spills into the accumulator file)
    int      every other VALU instruction (integer / logic / shifts / integer compares / conversions / lane reads)
    salu, lds, vmem, wait (s_waitcnt, s_nop), branch

Usage: python tools/inst_mix.py [extra -D flags ...] > profiles/<tag>_inst_mix.txt
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "optik_amd", "csrc")
sys.path.insert(0, ROOT)
from optik_amd import build as product_build  # noqa: E402

KERNEL = "_ZN5optik14ik_lane_kernelILi7ELb1EEEvNS_11SolveLaunchE"

# (file, first line, last line) -> region; the first match wins.  Line numbers follow the files as they are: the tool
# looks the markers up instead of hard-coding them.
MARKERS = {
    "ik_lane64.hpp": [
        ("struct Lane64Pipe", "hand-over (Lane64Pipe::event)"),
        ("template <int N, bool TIP>\nOPTIK_DEV void lane64_wave", "prologue / state init"),
        ("        // ---- refill: lanes without a restart pull the next work item", "refill"),
        ("        int32_t ret = 0;", "early-exit look + evaluation call"),
        ("        // ---- NLopt bookkeeping and Kraft's line search", "bookkeeping + BFGS call"),
        ("        // ---- labels 110/130: (reset,) search direction", "direction set-up (reset, LSQ call, records)"),
        ("            // ---- the wave's bounded problems by predicted class", "ranking"),
        ("            double y[2 * N];", "expand / read-back (records <-> blocks)"),
        ("            // ---- LDP tail (lsq_dual), back-substitution", "LDP tail + descent test"),
        ("        if (stepping && ret == 0 && !again) {", "trial point + publish"),
    ],
    "ik_nnls_quad.hpp": [
        ("template <int N, class Pipe = NoPipe, class Stop = NoStop>", "NNLS loop: set-up / control"),
        ("        // ---------------- steps two .. five", "NNLS steps 2-4 (duals, choice)"),
        ("                // step five: Householder construction", "NNLS step 5 (Householder + apply)"),
        ("        // ---------------- steps six .. ten", "NNLS steps 6-10 (solve, step length)"),
        ("        // ---------------- step eleven", "NNLS step 11 (Givens removal)"),
    ],
    "ik_slsqp.hpp": [
        ("OPTIK_DEV void rotg", "NNLS step 11 (Givens removal)"),
        ("template <int M>\nOPTIK_DEV double pick", "misc"),
        ("OPTIK_DEV int lsq_factor", "LSQ factor (E, f, Householder pass)"),
        ("OPTIK_DEV bool lsq_bound_rows", "rows of E^-1 + bound rows"),
        ("OPTIK_DEV void lsq_finish", "back-substitution (lsq_finish)"),
        ("OPTIK_DEV void ldl_update", "BFGS (ldl_update x 2)"),
        ("OPTIK_DEV void bfgs_update", "BFGS (ldl_update x 2)"),
    ],
}
WHOLE_FILE = {"ik_eval.hpp": "evaluation", "ik_math.hpp": "evaluation", "ik_nnls_first.hpp": "first NNLS pass (per lane)",
              "ik_solve.hpp": "refill (ChaCha seed, work counter) / stop tests", "ik_lane.hpp": "cross-lane moves",
              "ik_lane_kernel.hip": "prologue / state init", "ik_launch.hpp": "prologue / state init",
              "ik_platform.hpp": "misc"}
ORDER = ["refill", "refill (ChaCha seed, work counter) / stop tests", "early-exit look + evaluation call", "evaluation",
         "bookkeeping + BFGS call", "BFGS (ldl_update x 2)", "direction set-up (reset, LSQ call, records)",
         "LSQ factor (E, f, Householder pass)", "rows of E^-1 + bound rows", "first NNLS pass (per lane)", "ranking",
         "expand / read-back (records <-> blocks)", "hand-over (Lane64Pipe::event)", "NNLS loop: set-up / control",
         "NNLS steps 2-4 (duals, choice)", "NNLS step 5 (Householder + apply)", "NNLS steps 6-10 (solve, step length)",
         "NNLS step 11 (Givens removal)", "cross-lane moves", "LDP tail + descent test", "back-substitution (lsq_finish)",
         "trial point + publish", "prologue / state init", "misc", "?"]
CLASSES = ["f64", "cmp64", "select", "mov", "dpp", "agpr", "int", "salu", "lds", "vmem", "wait", "branch"]


def region_tables():
    tabs = {}
    for fn, marks in MARKERS.items():
        text = open(os.path.join(CSRC, fn)).read()
        rows = []
        for needle, region in marks:
            i = text.find(needle)
            if i < 0:
                raise SystemExit(f"{fn}: marker not found: {needle[:50]!r}")
            rows.append((text.count("\n", 0, i) + 1, region))
        tabs[fn] = sorted(rows)
    return tabs


def classify(op, rest):
    if op.startswith("v_accvgpr"):
        return "agpr"
    if op.startswith("v_"):
        if "quad_perm" in rest or "row_" in rest or "dpp" in op or op.startswith("v_permlane"):
            return "dpp"
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            return "cmp64" if op.endswith("_f64_e32") or op.endswith("_f64_e64") or "_f64" in op else "int"
        if "_f64" in op and not op.startswith("v_cvt"):
            return "f64"
        if op.startswith("v_cndmask"):
            return "select"
        if op.startswith("v_mov_b"):
            return "mov"
        return "int"
    if op in ("s_waitcnt", "s_nop", "s_sleep", "s_barrier") or op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_bpermute") or op.startswith("ds_swizzle") or op.startswith("ds_permute"):
        return "dpp"
    if op.startswith("ds_"):
        return "lds"
    if op.split("_")[0] in ("global", "flat", "buffer", "scratch"):
        return "vmem"
    return None


def compile_isa(extra, lines):
    unit = next(u for u in product_build.UNITS if u[1] == "ik_lane_kernel.o")
    out = "/tmp/inst_mix_%s.s" % ("g" if lines else "plain")
    cmd = ["/opt/rocm/bin/hipcc", *product_build.FLAGS, *unit[2], "-DOPTIK_LANE_ONLY_N=7", *extra, "-x", "hip", "--cuda-device-only", "-S",
           "ik_lane_kernel.hip", "-o", out] + (["-gline-tables-only"] if lines else [])
    subprocess.run(cmd, cwd=CSRC, check=True, capture_output=True)
    return open(out).read()


def kernel_body(text):
    i = text.index("\n" + KERNEL + ":")
    j = text.index(".Lfunc_end", i)
    return text[i:j]


def main():
    extra = sys.argv[1:]
    tabs = region_tables()
    plain = kernel_body(compile_isa(extra, False))
    n_plain = sum(1 for ln in plain.splitlines() if re.match(r"^\s+(v_|s_|ds_|global_|flat_|buffer_|scratch_)", ln))
    text = compile_isa(extra, True)
    files = {}
    for m in re.finditer(r'^\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', text, re.M):
        files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    body = kernel_body(text)
    counts = collections.defaultdict(lambda: collections.Counter())
    cur = "?"
    total = 0
    for ln in body.splitlines():
        m = re.match(r"^\s*\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            fn, line = files.get(int(m.group(1)), "?"), int(m.group(2))
            if fn in tabs:
                cur = "?"
                for first, region in tabs[fn]:
                    if line >= first:
                        cur = region
            else:
                cur = WHOLE_FILE.get(fn, "misc")
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\s*(.*)$", ln)
        if not m:
            continue
        cls = classify(m.group(1), m.group(2))
        if cls is None:
            continue
        counts[cur][cls] += 1
        total += 1
    print(f"# static instruction mix of ik_lane_kernel<7, true> ({' '.join(extra) or 'product flags'}); "
          f"{total} instructions with line tables, {n_plain} without")
    hdr = "%-46s" % "region" + "".join("%8s" % c for c in CLASSES) + "%8s%8s%9s" % ("VALU", "all", "non-f64")
    print(hdr)
    tot = collections.Counter()
    for region in ORDER + sorted(set(counts) - set(ORDER)):
        c = counts.get(region)
        if not c:
            continue
        valu = sum(c[k] for k in ("f64", "cmp64", "select", "mov", "dpp", "agpr", "int"))
        nf = 1.0 - c["f64"] / valu if valu else 0.0
        print("%-46s" % region + "".join("%8d" % c[k] for k in CLASSES) + "%8d%8d%8.0f%%" % (valu, sum(c.values()), 100 * nf))
        tot.update(c)
    valu = sum(tot[k] for k in ("f64", "cmp64", "select", "mov", "dpp", "agpr", "int"))
    print("%-46s" % "TOTAL" + "".join("%8d" % tot[k] for k in CLASSES) + "%8d%8d%8.0f%%" % (valu, sum(tot.values()), 100 * (1 - tot["f64"] / valu)))
    print("%-46s" % "share of VALU" + "".join(("%7.1f%%" % (100.0 * tot[k] / valu)) if k in ("f64", "cmp64", "select", "mov", "dpp", "agpr", "int") else "%8s" % "" for k in CLASSES))


if __name__ == "__main__":
    main()

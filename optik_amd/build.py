"""In-tree build of the gfx950 shared library (hipcc cross-compiles without a GPU).

``liboptik_amd.so`` = HIP kernels + the kernel-layer C ABI (include/optik_hip.h) +
the reference-compatible host API (include/optik.h, ``optik_robot_*``).
-ffp-contract=off is part of the numerical contract (see csrc/ik_math.hpp).
"""
from __future__ import annotations

import json
import os
import re
import shutil
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "liboptik_amd.so")
SOURCES = ["ik_capi.hip", "ik_select.hip", "ik_batch_ops.hip", "ik_quad_kernel.hip", "ik_lane_kernel.hip", "ik_wide_kernel.hip",
           "robot_host.cpp"]
# translation units: (source, object, extra flags).  ik_quad_kernel.hip is compiled twice -- its
# throughput form (two waves per SIMD) without the machine-LICM pass, which otherwise hoists constants and
# LDS addresses out of the solver loop only for the register allocator to spill them to scratch
# (csrc/ik_quad_kernel.hip), its latency forms and the launch function with the default pipeline.
UNITS = [("ik_capi.hip", "ik_capi.o", []),          # chains, options, the restart launch (host code only)
         ("ik_select.hip", "ik_select.o", []),      # selection kernels
         ("ik_batch_ops.hip", "ik_batch_ops.o", []),  # objective / FK / seed batches, probes
         # (the quad solver is issue-bound: the max-ILP scheduling strategy is worth +3 % restarts/s on the
         # throughput form and -3 % on a single ik()'s latency.  The iterative-ILP strategy the lane kernel is built
         # with: together with -disable-machine-licm it crashes this compiler on the quad kernel; without, it compiles
         # and is 3 % slower on both quad objects)
         # (the latency forms without the machine-LICM pass as well since round 4: 206 -> 201 us per single ik() call,
         # 673 -> 662 us deterministic, round 4: profiles/r4g_single_call.txt; in round 3 the default pipeline was the faster one there)
         ("ik_quad_kernel.hip", "ik_quad_latency.o",
          ["-DOPTIK_QUAD_PART=1", "-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
         ("ik_quad_kernel.hip", "ik_quad_throughput.o",
          ["-DOPTIK_QUAD_PART=2", "-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-use-amdgpu-trackers=1",
           "-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
         # the throughput form for n <= 7: one restart per lane, one wave per SIMD (ik_lane64.hpp)
         # (scheduling strategy, tools/ab_kernel_path.sh, M restarts/s interleaved on one box: the default 27.8, max-ILP 28.5,
         # iterative-ILP 29.2 (iterative-minreg 27.7, max-memory-clause 28.0); the register allocator assigning local
         # intervals in reverse order: +0.4 %, three of three runs -- 1 516 AGPR copies in the loop)
         # (-O2, overriding FLAGS' -O3: 29.48 -> 29.85 M, three of three runs; -O1 28.2, -Os 29.5)
         ("ik_lane_kernel.hip", "ik_lane_kernel.o", ["-O2", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp",
                                                     "-mllvm", "-greedy-reverse-local-assignment=1"]),
         # chains with 9 .. 16 joint positions: one run-time-n body per kernel (ik_wide.hpp)
         ("ik_wide_kernel.hip", "ik_wide_kernel.o", []),
         ("robot_host.cpp", "robot_host.o", [])]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-Wno-unused-value", "-pthread"]
# What a translation unit depends on is what the compiler says it read: every object is compiled with
# -MD and its dependency file (<object>.d) is kept next to it; an object is stale when any file listed
# there is newer than it, when the list is missing, or when it was compiled with other flags.  (Rounds
# 1-3 kept the header lists by hand, and they drifted.)


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _unit_cmd(hipcc, src, objname, extra):
    sp = os.path.join(CSRC, src)
    obj = os.path.join(CSRC, objname)
    # (-Rpass-analysis: a remark per kernel with its registers / scratch / LDS -- the toolchain guard below reads it)
    return [hipcc, *FLAGS, *extra, "-Rpass-analysis=kernel-resource-usage", "-MD", "-MF", obj + ".d", "-x", "hip", "-c", sp, "-o", obj]


# ---- toolchain-regression guard ------------------------------------------------------------------------------------
# The two throughput kernels live at the edge of the register file: ik_lane_kernel<7, *> uses
# 256 VGPR + 249 AGPR of the 512 a lone wave may have, ik_quad_kernel<7, *, 2> 255 of the 256 two waves per SIMD leave
# each.  A compiler update (or an innocent edit) that spills either to scratch costs a large part of the rate and nothing
# would say so.  Every compile therefore records what the compiler reports per kernel (<object>.res), and build() refuses
# a library whose kernels break the limits below (OPTIK_ALLOW_RESOURCE_REGRESSION=1: warns instead).
RESOURCE_LIMITS = [
    # (kernel-name regex, {field: maximum})
    (r"^optik::ik_lane_kernel<7, (true|false)>$", {"scratch": 0, "registers": 512}),
    (r"^optik::ik_quad_kernel<7, (true|false), 2>$", {"scratch": 0, "vgpr": 256}),
]


def _demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    except OSError:
        return list(names)
    return [(d or n) for n, d in zip(names, out + [""] * len(names))]


def parse_resource_remarks(stderr_text):
    """{kernel name: {vgpr, agpr, sgpr, scratch, lds, occupancy}} from -Rpass-analysis=kernel-resource-usage remarks."""
    rows, cur = [], None
    for ln in stderr_text.splitlines():
        m = re.search(r"remark: +(.*?) \[-Rpass", ln)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"mangled": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    names = _demangle([r["mangled"] for r in rows])
    out = {}
    for r, name in zip(rows, names):
        name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]

        def num(key):
            try:
                return int(r.get(key, "-1"))
            except ValueError:
                return -1
        out[name] = {"vgpr": num("VGPRs"), "agpr": num("AGPRs"), "sgpr": num("TotalSGPRs"),
                     "scratch": num("ScratchSize [bytes/lane]"), "lds": num("LDS Size [bytes/block]"),
                     "occupancy": num("Occupancy [waves/SIMD]")}
        out[name]["registers"] = max(out[name]["vgpr"], 0) + max(out[name]["agpr"], 0)
    return out


def kernel_resources():
    """What the compiler reported for every kernel of the library as built ({} for units compiled before the guard)."""
    res = {}
    for _, objname, _ in UNITS:
        try:
            with open(os.path.join(CSRC, objname + ".res")) as fh:
                res.update(json.load(fh))
        except (OSError, ValueError):
            pass
    return res


def check_resources(resources=None, limits=None):
    """Violations of RESOURCE_LIMITS as a list of strings; a guarded kernel that is missing from the record is one too."""
    resources = kernel_resources() if resources is None else resources
    bad = []
    for pat, lim in (RESOURCE_LIMITS if limits is None else limits):
        hits = [k for k in resources if re.match(pat, k)]
        if not hits:
            bad.append(f"no kernel matching {pat} in the compiler's resource report")
        for k in hits:
            for field, mx in lim.items():
                if resources[k].get(field, -1) > mx or resources[k].get(field, -1) < 0:
                    bad.append(f"{k}: {field} = {resources[k].get(field)} (limit {mx}); all: {resources[k]}")
    return bad


def toolchain_version(hipcc=None):
    """First lines of `hipcc --version` (HIP version + clang version): what profiles/perf_floor.json is keyed by."""
    try:
        out = subprocess.run([hipcc or _hipcc(), "--version"], capture_output=True, text=True).stdout.splitlines()
    except (OSError, RuntimeError):
        return None
    return " | ".join(ln.strip() for ln in out[:2])


def _dep_files(obj):
    """Prerequisites recorded by -MD for `obj` (None: no record)."""
    try:
        text = open(obj + ".d").read()
    except OSError:
        return None
    text = text.replace("\\\n", " ")
    if ":" not in text:
        return None
    return [t for t in text.split(":", 1)[1].split() if t]


def _unit_stale(hipcc, src, objname, extra):
    obj = os.path.join(CSRC, objname)
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    flags_file = obj + ".flags"
    if hipcc is not None and not (os.path.exists(flags_file)
                                  and open(flags_file).read() == " ".join(_unit_cmd(hipcc, src, objname, extra)[1:])):
        return True
    deps = _dep_files(obj)
    if deps is None:
        return True
    for d in deps:
        d = d if os.path.isabs(d) else os.path.join(CSRC, d)
        if not os.path.exists(d) or os.path.getmtime(d) > t:
            return True
    return False


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    # (a runtime-only machine has the library and no compiler: the recorded command lines then cannot be
    # compared -- the prerequisites' time stamps still are)
    try:
        hipcc = _hipcc()
    except RuntimeError:
        hipcc = None
    t = os.path.getmtime(LIB)
    for src, objname, extra in UNITS:
        if _unit_stale(hipcc, src, objname, extra) or os.path.getmtime(os.path.join(CSRC, objname)) > t:
            return True
    return os.path.getmtime(os.path.abspath(__file__)) > t


def _compile(job):
    cmd, verbose = job
    if verbose:
        print(" ".join(cmd), flush=True)
    t0 = time.perf_counter()
    p = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
    # (the compiler's own diagnostics, without the resource remarks and their source excerpts)
    noise = [ln for ln in p.stderr.splitlines()
             if not ("-Rpass-analysis=kernel-resource-usage" in ln or re.match(r"^\s*\d* *\|", ln) or "remark" in ln)]
    if p.returncode != 0:
        sys.stderr.write(p.stderr)
        raise subprocess.CalledProcessError(p.returncode, cmd)
    if noise and verbose:
        sys.stderr.write("\n".join(noise) + "\n")
    with open(cmd[-1] + ".res", "w") as fh:
        json.dump(parse_resource_remarks(p.stderr), fh, indent=0, sort_keys=True)
    with open(cmd[-1] + ".flags", "w") as fh:
        fh.write(" ".join(cmd[1:]))
    return cmd[-1], time.perf_counter() - t0


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> str:
    """Compiles the stale translation units (all of them with force) IN PARALLEL and links the library."""
    if not force and not is_stale():
        return LIB
    hipcc = _hipcc()
    todo = [(_unit_cmd(hipcc, src, objname, extra), verbose) for src, objname, extra in UNITS
            if force or _unit_stale(hipcc, src, objname, extra)]
    t0 = time.perf_counter()
    if todo:
        workers = max(1, min(len(todo), jobs or (os.cpu_count() or 1)))
        with ThreadPoolExecutor(workers) as ex:
            for obj, dt in ex.map(_compile, todo):
                if verbose:
                    print(f"  {os.path.basename(obj)}: {dt:.1f} s", flush=True)
    bad = check_resources()
    if bad:
        msg = ("toolchain-regression guard (optik_amd/build.py: RESOURCE_LIMITS) -- " + toolchain_version(hipcc) + ":\n  "
               + "\n  ".join(bad))
        if os.environ.get("OPTIK_ALLOW_RESOURCE_REGRESSION") == "1":
            sys.stderr.write("WARNING: " + msg + "\n")
        else:
            raise RuntimeError(msg + "\n(OPTIK_ALLOW_RESOURCE_REGRESSION=1 links the library anyway)")
    objs = [os.path.join(CSRC, objname) for _, objname, _ in UNITS]
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    with open(os.path.join(CSRC, "toolchain.json"), "w") as fh:
        json.dump({"hipcc": toolchain_version(hipcc), "built": time.strftime("%Y-%m-%d %H:%M:%S"),
                   "guarded_kernels": {k: v for k, v in kernel_resources().items()
                                       if any(re.match(p, k) for p, _ in RESOURCE_LIMITS)}}, fh, indent=1)
    if verbose:
        print(f"built {len(todo)} of {len(UNITS)} units + link in {time.perf_counter() - t0:.1f} s "
              f"({'forced' if force else 'stale units only'})")
    return LIB


def print_resources():
    """Registers / scratch / LDS / occupancy of every kernel of the library as built (what the compiler reported)."""
    print("%-44s %5s %5s %5s %9s %7s %10s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch B", "LDS B", "waves/SIMD"))
    for name, r in sorted(kernel_resources().items()):
        print("%-44s %5d %5d %5d %9d %7d %10d" % (name, r["vgpr"], r["agpr"], r["sgpr"], r["scratch"], r["lds"], r["occupancy"]))
    print("# " + str(toolchain_version()))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--resources" in sys.argv:
        print_resources()

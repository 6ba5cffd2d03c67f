#!/usr/bin/env python3
"""Builds optik_amd/csrc/variants/<name>.so: the library with every translation unit compiled with the
product's own flags (optik_amd/build.py:UNITS) plus extra defines -- e.g. the -DOPTIK_PROFILE build that
phase_profile.py (a rounds 3-5 tool: git history) loads through OPTIK_PROF_LIB.  Units are compiled in parallel; objects are kept under
variants/<name>/ and reused while their flags and sources are unchanged.  hipcc cross-compiles without a GPU.

Usage: python tools/build_lib_variant.py <name> [-DFLAG ...] [--only=object.o[,object.o]]
(--only: just these objects get the extra flags, the others are the product build's own objects)"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optik_amd import build as pb  # noqa: E402


def main():
    name = sys.argv[1]
    only = [a.split("=", 1)[1].split(",") for a in sys.argv[2:] if a.startswith("--only=")]
    only = only[0] if only else None
    extra = [a for a in sys.argv[2:] if a.startswith("-") and not a.startswith("--only=")]
    vdir = os.path.join(pb.CSRC, "variants", name)
    os.makedirs(vdir, exist_ok=True)
    hipcc = pb._hipcc()
    jobs, objs = [], []
    for src, objname, uextra in pb.UNITS:
        if only is not None and objname not in only:
            objs.append(os.path.join(pb.CSRC, objname))
            continue
        obj = os.path.join(vdir, objname)
        objs.append(obj)
        # units the extra defines cannot change are taken from the product build as they are
        cmd = [hipcc, *pb.FLAGS, *uextra, *extra, "-x", "hip", "-c", os.path.join(pb.CSRC, src), "-o", obj]
        stamp = obj + ".cmd"
        srcs = [os.path.join(pb.CSRC, f) for f in os.listdir(pb.CSRC) if f.endswith((".hpp", ".hip", ".cpp"))]
        fresh = (os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
                 and all(os.path.getmtime(s) <= os.path.getmtime(obj) for s in srcs))
        if not fresh:
            jobs.append((cmd, stamp))

    def run(job):
        cmd, stamp = job
        subprocess.check_call(cmd, cwd=pb.CSRC, stderr=subprocess.DEVNULL)
        open(stamp, "w").write(" ".join(cmd))

    with ThreadPoolExecutor(max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    out = os.path.join(pb.CSRC, "variants", name + ".so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", *objs, "-o", out])
    print(out)


if __name__ == "__main__":
    main()

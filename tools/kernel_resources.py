#!/usr/bin/env python3
"""Compiler-reported resources of every gfx950 kernel (registers, scratch, LDS, occupancy):
hipcc -Rpass-analysis=kernel-resource-usage on every HIP translation unit of optik_amd/build.py with its flags.
Runs without a GPU.  Usage: python tools/kernel_resources.py > profiles/<tag>_kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "optik_amd", "csrc")
sys.path.insert(0, ROOT)
from optik_amd import build as product_build  # noqa: E402  (the translation units and their flags)

out = ""
for src, objname, extra in product_build.UNITS:
    if not src.endswith(".hip"):
        continue
    cmd = ["/opt/rocm/bin/hipcc", *product_build.FLAGS, *extra, "-x", "hip", "-c", src, "-o", "/tmp/" + objname + ".rpass.o",
           "-Rpass-analysis=kernel-resource-usage"]
    out += subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
rows, cur = [], None
for ln in out.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", ln)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        except OSError:
            dem = ""
        name = (dem or name).replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        cur = {"name": name}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print("%-40s %5s %5s %5s %9s %7s %10s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch B", "LDS B", "waves/SIMD"))
for r in sorted(rows, key=lambda r: r["name"]):
    print("%-40s %5s %5s %5s %9s %7s %10s" % (
        r["name"], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", "?"),
        r.get("ScratchSize [bytes/lane]", "?"), r.get("LDS Size [bytes/block]", "?"),
        r.get("Occupancy [waves/SIMD]", "?")))

#!/usr/bin/env python3
"""Folds the rocprofv3 passes of tools/profile_round.sh into one record for the bench command:

  kernels[name] = FETCH_SIZE / WRITE_SIZE KB per launch (separate passes), launches, and from the SQ
                  pass the f64 instruction mix and VALU-busy fraction -- all over the launches of
                  the TIMED run only (everything after the last eng_init_kernel dispatch: the
                  warm-up run's launches are a different population)
  f64_flops_per_restart = (ADD_F64 + MUL_F64 + 2 FMA_F64 + TRANS_F64) x 64 lanes over those
                  launches / restarts of the timed run; wave-level counts (EXEC masks are not
                  applied: an upper bound on lane flops), scaled by grid waves / SQ_WAVES when a
                  pass sees only part of a kernel's waves

Writes <out>/pmc_by_command.json = {"key": ..., "record": ...}; profiles/r2_pmc_by_command.json
collects such records under "commands".
"""
import collections
import csv
import glob
import json
import re
import sys

out, cmd = sys.argv[1], sys.argv[2]
line = json.loads([ln for ln in open(f"{out}/bench_unprofiled.json") if ln.startswith('{"metric"')][0])
KNAME = re.compile(r"(eng_[a-z_]+_kernel|ik_solve_kernel|ik_tile_argmin_kernel|ik_select_kernel)")


def rows(d):
    f = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)[0]
    rs = list(csv.DictReader(open(f)))
    # dispatches are numbered in launch order; the timed run starts at the last eng_init_kernel
    init = max((int(r["Dispatch_Id"]) for r in rs if "eng_init_kernel" in r["Kernel_Name"]), default=0)
    return [r for r in rs if int(r["Dispatch_Id"]) >= init]


rec = {"command": cmd, "kernels": {}}
for name, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    acc = collections.defaultdict(lambda: [0.0, set()])
    for r in rows(d):
        m = KNAME.search(r["Kernel_Name"])
        if m and r["Counter_Name"] == name:
            acc[m.group(1)][0] += float(r["Counter_Value"])
            acc[m.group(1)][1].add(r["Dispatch_Id"])
    for k, (v, ids) in acc.items():
        e = rec["kernels"].setdefault(k, {})
        e[name + "_kb_per_launch"] = v / max(len(ids), 1)
        e["launches"] = len(ids)
# SQ pass
per = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for r in rows("sq"):
    m = KNAME.search(r["Kernel_Name"])
    if not m:
        continue
    did = r["Dispatch_Id"]
    per[did][r["Counter_Name"]] += float(r["Counter_Value"])
    meta[did] = (m.group(1), int(r["Grid_Size"]) // 64 if "Grid_Size" in r else 0,
                 (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 if "End_Timestamp" in r else 0.0)
flops_total = 0.0
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for did, c in per.items():
    k, grid_waves, dur_us = meta[did]
    seen = c.get("SQ_WAVES", 0.0)
    scale = (grid_waves / seen) if (seen > 0 and grid_waves > 0) else 1.0
    fl = (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2.0 * c["SQ_INSTS_VALU_FMA_F64"]
          + c["SQ_INSTS_VALU_TRANS_F64"]) * 64.0 * scale
    flops_total += fl
    a = agg[k]
    a["f64_flops"] += fl
    a["valu_insts"] += c["SQ_INSTS_VALU"] * scale
    a["active_quad_cycles"] += c["SQ_ACTIVE_INST_VALU"] * scale
    a["dur_us"] += dur_us
    a["n"] += 1
valu_busy = {}
for k, a in agg.items():
    e = rec["kernels"].setdefault(k, {})
    e["f64_flops_per_launch"] = a["f64_flops"] / a["n"]
    e["valu_insts_per_launch"] = a["valu_insts"] / a["n"]
    # quad-cycles x 4 / (1024 SIMDs x kernel time x 2.4 GHz); kernels of other sub-pools run beside it
    if a["dur_us"] > 0:
        valu_busy[k] = a["active_quad_cycles"] * 4.0 / (1024.0 * a["dur_us"] * 2400.0)
restarts = line["config"]["restarts_per_gpu"] * line["steps"] * line["n_gpus"]
rec["f64_flops_per_restart"] = flops_total / restarts if restarts else None
rec["valu_busy"] = valu_busy
rec["unprofiled_value"] = line["value"]
key = line["config"]["command_key"]
json.dump({"key": key, "record": rec, "bench_line": line}, open(f"{out}/pmc_by_command.json", "w"), indent=1)
# the collection bench.py reads (copied from gpurun_out/ to profiles/ after the run)
import os
coll_path = "gpurun_out/r2_pmc_by_command.json"
coll = {"commands": {}}
for cand in (coll_path, "profiles/r2_pmc_by_command.json"):
    if os.path.exists(cand):
        coll = json.load(open(cand))
        break
coll.setdefault("commands", {})[key] = rec
coll["note"] = ("rocprofv3 PMC passes per bench command (tools/profile_round.sh): FETCH_SIZE / WRITE_SIZE in KB per launch "
                "of the timed run, separate passes; SQ pass for the f64 instruction mix and VALU-busy")
json.dump(coll, open(coll_path, "w"), indent=1)
print(json.dumps({k: {kk: (round(vv, 1) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                  for k, v in rec["kernels"].items()}, indent=1))
print("f64 flops per restart", rec["f64_flops_per_restart"], "valu_busy", valu_busy)

#!/usr/bin/env python3
"""Folds the rocprofv3 passes of tools/profile_kernel_path.sh into one record for the solve kernel of
`bench.py`: over the launches of the TIMED repetitions (the last `reps` launches of the
kernel; the warm-up launch has a different size), per launch and per restart:

  FETCH_SIZE / WRITE_SIZE (KB, separate passes) -> HBM bytes per restart = (2 FETCH + WRITE) x 1024 /
      restarts per launch (FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950)
  SQ pass: f64 instruction mix -> flops per restart (wave-level counts x 64 lanes: an upper bound),
      VALU-busy = SQ_ACTIVE_INST_VALU x 4 / SQ_WAVE_CYCLES... reported as the ratio of the two counters
Writes <out>/pmc_kernel_path.json."""
import collections
import csv
import glob
import json
import sys

out, cmd = sys.argv[1], sys.argv[2]
line = json.loads([ln for ln in open(f"{out}/bench_unprofiled.json") if ln.startswith('{"metric"')][0])
reps = line["config"]["reps"]
per_launch = line["roofline"]["units_per_launch"]
KERNELS = ("ik_quad_kernel", "ik_lane_kernel", "wide_solve_kernel", "wide_solve_coop_kernel")


def timed_rows(d):
    f = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)[0]
    rs = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in KERNELS)]
    ids = sorted({int(r["Dispatch_Id"]) for r in rs})[-reps:]
    return [r for r in rs if int(r["Dispatch_Id"]) in ids], len(ids)


rec = {"command": cmd, "key": line["config"]["command_key"], "restarts_per_launch": per_launch, "launches": reps}
for name, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    rs, n = timed_rows(d)
    tot = sum(float(r["Counter_Value"]) for r in rs if r["Counter_Name"] == name)
    rec[name + "_kb_per_launch"] = tot / max(n, 1)
rec["hbm_bytes_per_restart"] = (2.0 * rec["FETCH_SIZE_kb_per_launch"] + rec["WRITE_SIZE_kb_per_launch"]) * 1024.0 / per_launch
rs, n = timed_rows("sq")
acc = collections.defaultdict(float)
dur = collections.defaultdict(float)
for r in rs:
    acc[r["Counter_Name"]] += float(r["Counter_Value"])
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
grid_waves = [int(r["Grid_Size"]) // 64 for r in rs if "Grid_Size" in r][:1]
scale = 1.0
if grid_waves and acc.get("SQ_WAVES"):
    scale = grid_waves[0] * n / acc["SQ_WAVES"]  # a pass may see only part of a kernel's waves
flops = (acc["SQ_INSTS_VALU_ADD_F64"] + acc["SQ_INSTS_VALU_MUL_F64"] + 2.0 * acc["SQ_INSTS_VALU_FMA_F64"]
         + acc["SQ_INSTS_VALU_TRANS_F64"]) * 64.0 * scale
rec["f64_flops_per_restart"] = flops / (per_launch * max(n, 1))
rec["valu_insts_per_restart"] = acc["SQ_INSTS_VALU"] * 64.0 * scale / (per_launch * max(n, 1))
rec["valu_busy"] = acc["SQ_ACTIVE_INST_VALU"] / acc["SQ_WAVE_CYCLES"] if acc.get("SQ_WAVE_CYCLES") else None
rec["sq_scale"] = scale
try:
    rs2, n2 = timed_rows("sq2")
    a2 = collections.defaultdict(float)
    for r in rs2:
        a2[r["Counter_Name"]] += float(r["Counter_Value"])
    if a2.get("SQ_ACTIVE_INST_VALU"):
        # lanes active per VALU instruction cycle (EXEC masks applied): what part of the wave-level flops are lane flops
        rec["valu_active_lane_frac"] = a2["SQ_THREAD_CYCLES_VALU"] / (64.0 * a2["SQ_ACTIVE_INST_VALU"])
    if a2.get("SQ_WAVE_CYCLES"):
        rec["wait_any_frac"] = a2["SQ_WAIT_ANY"] / a2["SQ_WAVE_CYCLES"]
        rec["wait_inst_any_frac"] = a2["SQ_WAIT_INST_ANY"] / a2["SQ_WAVE_CYCLES"]
        rec["lds_insts_per_restart"] = a2["SQ_INSTS_LDS"] * 64.0 / (per_launch * max(n2, 1)) / 64.0
except (IndexError, OSError, KeyError):
    pass
rec["kernel_ms_under_profiler"] = sum(dur.values()) / max(len(dur), 1)
rec["bench_value_unprofiled"] = line["value"]
json.dump(rec, open(f"{out}/pmc_kernel_path.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
# ... and into the round's by-command file (what bench.py reads roofline.traffic and the VALU scalars from): copy
# gpurun_out/r6_pmc_by_command.json to profiles/ to make it the record of that command
by_cmd = "gpurun_out/r6_pmc_by_command.json"
try:
    doc = json.load(open(by_cmd))
except (OSError, ValueError):
    doc = {"note": "PMC records of bench.py commands (tools/profile_kernel_path.sh: FETCH_SIZE, WRITE_SIZE and two SQ passes, "
                   "each its own rocprofv3 run with --kernel-trace only), keyed by bench.py's command key", "commands": {}}
doc["commands"][rec["key"]] = {"kernel_path": rec}
json.dump(doc, open(by_cmd, "w"), indent=1)

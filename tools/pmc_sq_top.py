#!/usr/bin/env python3
"""Steady-state view of a tools/pmc_sq.sh pass: median counters of the 100 busiest launches per kernel."""
import collections
import csv
import glob
import statistics
import sys

f = glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
by = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    for nm in ("nnls", "eval", "update", "finish", "bucket"):
        if "eng_" + nm in r["Kernel_Name"]:
            k = (int(r["Dispatch_Id"]), nm, r["Grid_Size"])
            by[k][r["Counter_Name"]] = float(r["Counter_Value"])
            by[k]["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for nm in ("eval", "update", "bucket", "nnls", "finish"):
    ks = [k for k in by if k[1] == nm]
    if not ks:
        continue
    ks.sort(key=lambda k: -by[k].get("SQ_INSTS_VALU", 0))
    top = ks[:100]
    med = {c: statistics.median(by[k][c] for k in top) for c in by[top[0]]}
    line = {c: round(v) for c, v in med.items()}
    if "SQ_WAVE_CYCLES" in med and med.get("dur_us"):
        # quad-cycles -> resident waves per SIMD at 2.4 GHz, 1024 SIMDs
        line["resident_waves_per_simd@2.4GHz"] = round(med["SQ_WAVE_CYCLES"] * 4 / 1024 / (med["dur_us"] * 2400), 2)
        line["valu_busy@2.4GHz"] = round(med["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (med["dur_us"] * 2400), 2)
    print(nm, top[0][2], line)

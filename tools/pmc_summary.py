#!/usr/bin/env python3
"""Per-kernel mean FETCH_SIZE / WRITE_SIZE per launch from the two rocprofv3 PMC passes."""
import collections
import csv
import glob
import json
import re
import sys

out, cmd = sys.argv[1], sys.argv[2]
res = {"command": cmd, "unit": "KB per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes)",
       "kernels": {}}
for name, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = glob.glob(f"{out}/{d}/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        m = re.search(r"(eng_[a-z_]+_kernel|ik_solve_kernel|ik_tile_argmin_kernel|ik_select_kernel)",
                      r["Kernel_Name"])
        if m:
            acc[m.group(1)][0] += float(r["Counter_Value"])
            acc[m.group(1)][1] += 1
    for k, v in acc.items():
        res["kernels"].setdefault(k, {})[name] = {"launches": v[1], "mean_kb_per_launch": v[0] / max(v[1], 1)}
json.dump(res, open(f"{out}/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))

#!/usr/bin/env python3
"""Per-kernel mean SQ counters per launch (and per wave) from a rocprofv3 --pmc pass."""
import collections
import csv
import glob
import json
import re
import sys

out = sys.argv[1]
f = glob.glob(f"{out}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    m = re.search(r"(eng_[a-z_]+_kernel|ik_solve_kernel)", r["Kernel_Name"])
    if not m:
        continue
    k = m.group(1)
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
res = {}
for k, c in acc.items():
    n = len(cnt[k])
    waves = c.get("SQ_WAVES", 0.0) or 1.0
    res[k] = {"launches": n}
    for name, v in c.items():
        res[k][name + "_per_launch"] = v / n
        if name != "SQ_WAVES":
            res[k][name + "_per_wave"] = v / waves
json.dump(res, open(f"{out}/sq_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))

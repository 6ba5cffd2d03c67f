#!/usr/bin/env python3
"""Per-stream view of a rocprofv3 --kernel-trace CSV of bench.py (engine path): for the timed run
(everything after the last eng_init_kernel) kernel totals, how many kernels run concurrently, and
the idle gaps of each sub-pool's stream by the pair of kernels around them.
Usage: python tools/trace_gaps.py <kernel_trace.csv> [window_start_ms window_end_ms]"""
import collections
import csv
import re
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
short = lambda n: (re.search(r"(eng_[a-z_]+|ik_[a-z_]+)", n) or re.search(r"(.{0,30})", n)).group(1)
inits = [i for i, r in enumerate(rows) if "eng_init_kernel" in r["Kernel_Name"]]
run = rows[inits[-1]:]
t0 = int(run[0]["Start_Timestamp"])
ev = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, short(r["Kernel_Name"]), r["Stream_Id"]) for r in run
      if "eng_" in r["Kernel_Name"]]
A = float(sys.argv[2]) * 1e6 if len(sys.argv) > 3 else 0.0
B = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else max(e for _, e, _, _ in ev)
sel = [x for x in ev if x[0] >= A and x[1] <= B]
print(f"window {A / 1e6:.1f} .. {B / 1e6:.1f} ms of the timed run")
by = collections.defaultdict(lambda: [0, 0])
for s, e, k, _ in sel:
    by[k][0] += e - s
    by[k][1] += 1
tot = sum(v[0] for v in by.values())
for k, (d, c) in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:26s} {d / 1e6:8.3f} ms  n {c:5d}  mean {d / c / 1e3:7.1f} us")
print(f"  kernel time / wall = {tot / (B - A):.2f} kernels running on average")
edges = sorted([(s, 1) for s, _, _, _ in sel] + [(e, -1) for _, e, _, _ in sel])
cur, last, hist = 0, A, collections.defaultdict(int)
for t, d in edges:
    hist[cur] += t - last
    last, cur = t, cur + d
print("  ms with k kernels running:", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
for sid in sorted({x[3] for x in sel}):
    v = sorted(x for x in sel if x[3] == sid)
    gaps = collections.defaultdict(list)
    for a, b in zip(v, v[1:]):
        gaps[a[2].replace("eng_", "").replace("_kernel", "") + "->" + b[2].replace("eng_", "").replace("_kernel", "")].append(b[0] - a[1])
    busy = sum(e - s for s, e, _, _ in v)
    print(f"  stream {sid}: busy {busy / 1e6:.2f} ms; idle gaps (count, mean us, max us):",
          {k: (len(g), round(statistics.mean(g) / 1e3, 1), round(max(g) / 1e3)) for k, g in gaps.items() if statistics.mean(g) > 2000})

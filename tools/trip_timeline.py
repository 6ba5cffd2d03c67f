#!/usr/bin/env python3
"""Trip timeline of the timed engine run in a rocprofv3 --kernel-trace CSV of bench.py: per
sub-pool stream, when each trip's update kernel starts and how long the trip takes, in 5 ms
buckets -- shows the fill phase, the steady state and the drain of a run.
Usage: python tools/trip_timeline.py <kernel_trace.csv> [bucket_ms]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
bucket = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
inits = [i for i, r in enumerate(rows) if "eng_init_kernel" in r["Kernel_Name"]]
run = rows[inits[-1]:]
t0 = int(run[0]["Start_Timestamp"])
ev = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"], r["Stream_Id"])
            for r in run if "eng_" in r["Kernel_Name"])
end = max(e for _, e, _, _ in ev)
print(f"timed run: {end / 1e6:.2f} ms of engine kernels")
for name in ("eng_tail_list_kernel", "eng_tail_quad_kernel"):
    for s, e, k, _ in ev:
        if name in k:
            print(f"  {name}: starts {s / 1e6:.2f} ms, runs {(e - s) / 1e6:.2f} ms")
for sid in sorted({x[3] for x in ev}):
    ups = [s for s, _, k, st in ev if st == sid and "eng_update_kernel" in k]
    if len(ups) < 3:
        continue
    per = collections.defaultdict(list)
    for a, b in zip(ups, ups[1:]):
        per[int(a / 1e6 / bucket)].append((b - a) / 1e3)
    print(f"  stream {sid}: {len(ups)} trips; mean trip time (us) per {bucket:g} ms bucket:")
    print("    " + "  ".join(f"{int(k * bucket)}:{sum(v) / len(v):.0f}x{len(v)}" for k, v in sorted(per.items())))
busy = collections.defaultdict(float)
for s, e, k, _ in ev:
    b0, b1 = int(s / 1e6 / bucket), int(e / 1e6 / bucket)
    for b in range(b0, b1 + 1):
        lo, hi = max(s, b * bucket * 1e6), min(e, (b + 1) * bucket * 1e6)
        busy[b] += max(0.0, hi - lo)
print("  kernels running on average per bucket: " + "  ".join(f"{int(b * bucket)}:{v / (bucket * 1e6):.2f}" for b, v in sorted(busy.items())))

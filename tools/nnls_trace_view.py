#!/usr/bin/env python3
"""Timeline summary of an OPTIK_NNLS_TRACE dump: per-wave start/end (100 MHz wall clock),
hardware placement and pass counts of one NNLS launch."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
st = (a[:, 0] - t0).astype(np.float64) / 100.0  # us
en = (a[:, 1] - t0).astype(np.float64) / 100.0
hw = a[:, 2] & 0xFFFFFFFF
xcc = ((a[:, 2] >> 32) & 0xFF).astype(np.int64)
passes = (a[:, 3] & 0xFFFF).astype(np.int16).astype(np.int64)
cyc = (a[:, 3] >> 16).astype(np.float64)
work = passes >= 0
print(f"waves {len(a)}  with work {work.sum()}  span {en.max():.1f} us")
d = en - st
print(f"shader clock while waves run: {np.median(cyc[d > 5] / d[d > 5]) / 1e3:.3f} GHz (s_memtime ticks per wall us)")
print(f"duration of working waves: mean {d[work].mean():.2f} us  p50 {np.median(d[work]):.2f}  p99 {np.percentile(d[work], 99):.2f}  max {d[work].max():.2f}")
for p in range(0, passes.max() + 1):
    m = passes == p
    if m.sum():
        print(f"  max passes {p:2d}: {m.sum():5d} waves  mean dur {d[m].mean():6.2f} us  start {st[m].min():6.1f}..{st[m].max():6.1f}")
# concurrency over time
ts = np.linspace(0, en.max(), 41)
for t in ts[:-1]:
    running = ((st <= t) & (en > t) & work).sum()
    print(f"  t={t:6.1f} us  running working waves {running}")
# placement: CU/SIMD ids from HW_ID (gfx9: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13])
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
se = (hw >> 13) & 7
key = xcc * 100000 + se * 1000 + cu * 10 + simd
u, c = np.unique(key[work], return_counts=True)
print(f"distinct (xcc,se,cu,simd) used {len(u)}  waves per SIMD: min {c.min()} mean {c.mean():.2f} max {c.max()}")
busy = np.zeros(len(u))
idx = {k: i for i, k in enumerate(u)}
for k, dd in zip(key[work], d[work]):
    busy[idx[k]] += dd
print(f"busy wave-us per SIMD: min {busy.min():.1f} mean {busy.mean():.1f} max {busy.max():.1f}  (span {en.max():.1f}; 2 slots per SIMD)")

#!/bin/bash
# Kernel timeline of single optik_robot_ik calls (tools/single_ik_latency.c under rocprofv3 --kernel-trace):
# per kernel name the mean duration, and the mean gap between consecutive kernels of a call.
#   tools/single_call_trace.sh [calls] [parallelism]   (on the GPU box; writes gpurun_out/single_trace/)
set -e
CALLS=${1:-300}
PAR=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/single_trace${PAR:+_p$PAR}
mkdir -p "$OUT"
gcc -O2 -std=c11 -I"$ROOT/include" "$ROOT/tools/single_ik_latency.c" -L"$ROOT/optik_amd/csrc" -loptik_amd \
    -Wl,-rpath,"$ROOT/optik_amd/csrc" -lm -o /tmp/lat
/tmp/lat "$ROOT/optik_amd/robots/panda.urdf" panda_link0 panda_link8 "$CALLS" $PAR | tee "$OUT/untraced.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -o t -- /tmp/lat "$ROOT/optik_amd/robots/panda.urdf" \
    panda_link0 panda_link8 "$CALLS" $PAR > "$OUT/traced.txt" 2>&1 || true
CSV=$(find "$OUT/raw" -name '*kernel_trace.csv' | head -1)
python3 - "$CSV" <<'PY' | tee "$OUT/summary.txt"
import collections, csv, re, statistics, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: re.sub(r"<.*", "", re.sub(r"^void (optik::)?", "", n))[:40]
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gap[short(a["Kernel_Name"]) + " -> " + short(b["Kernel_Name"])].append(g)
for r in rows:
    dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("kernel                                      n     mean us   median    p90")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k:40s} {len(v):6d} {statistics.mean(v)/1e3:9.1f} {v2[len(v2)//2]/1e3:9.1f} {v2[int(len(v2)*0.9)]/1e3:9.1f}")
print("gaps (mean us, median us, n):")
for k, v in sorted(gap.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 20:
        v2 = sorted(v)
        print(f"  {k:70s} {statistics.mean(v)/1e3:8.1f} {v2[len(v2)//2]/1e3:8.1f} {len(v):6d}")
PY
rm -rf "$OUT/raw"

#!/bin/bash
# SQ / instruction-cache counters of the solve kernel of single optik_robot_ik calls (tools/single_ik_latency.c),
# two PMC passes.  Usage (GPU box): tools/pmc_single_call.sh [calls] [parallelism]; output gpurun_out/pmc_single[_pN]/
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CALLS=${1:-200}
PAR=${2:-}
OUT=$ROOT/gpurun_out/pmc_single${PAR:+_p$PAR}
mkdir -p "$OUT"
gcc -O2 -std=c11 -I"$ROOT/include" "$ROOT/tools/single_ik_latency.c" -L"$ROOT/optik_amd/csrc" -loptik_amd \
    -Wl,-rpath,"$ROOT/optik_amd/csrc" -lm -o /tmp/lat
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
P2="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P3="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/p$i" -o t -- /tmp/lat "$ROOT/optik_amd/robots/panda.urdf" \
      panda_link0 panda_link8 "$CALLS" $PAR > "$OUT/p$i.txt" 2>&1
done
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import collections, csv, glob, sys
val = {}; calls = 0
for d in sorted(glob.glob(f"{sys.argv[1]}/p[0-9]")):
    tot = collections.defaultdict(float); seen = set()
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ik_quad_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); seen.add(r["Dispatch_Id"])
    calls = max(calls, len(seen))
    for k, v in tot.items(): val.setdefault(k, v / max(len(seen), 1))   # per launch; a counter of several passes: the first
print(f"solve-kernel launches per pass: {calls}; per launch:")
for k, v in sorted(val.items()): print(f"  {k:32s} {v:14.1f}")
g = lambda k: val.get(k, 0.0)
w = max(g("SQ_WAVES"), 1.0)
if g("SQC_ICACHE_REQ"): print("instruction cache: hit rate %.4f, %.1f misses per wave" % (g("SQC_ICACHE_HITS") / g("SQC_ICACHE_REQ"), g("SQC_ICACHE_MISSES") / w))
if g("SQ_WAVE_CYCLES"):
    cyc = 4.0 * g("SQ_WAVE_CYCLES") / w   # (SQ_WAVE_CYCLES, SQ_WAIT_* count quad-cycles)
    print("per wave: %.0f VALU + %.0f SALU + %.0f LDS + %.0f VMEM instructions in %.0f cycles = %.2f cycles per VALU instruction; "
          "parked at s_waitcnt %.3f, issue stalls %.3f of the wave's cycles" % (
          g("SQ_INSTS_VALU") / w, g("SQ_INSTS_SALU") / w, g("SQ_INSTS_LDS") / w, (g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR")) / w, cyc,
          cyc / max(g("SQ_INSTS_VALU") / w, 1.0), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")))
PY
rm -rf "$OUT"/p1 "$OUT"/p2 "$OUT"/p3

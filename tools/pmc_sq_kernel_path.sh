#!/bin/bash
# SQ wait / issue counters of the solve kernel of `bench.py --path kernel` (one PMC pass, 8 SQ slots).
# Usage: bash tools/pmc_sq_kernel_path.sh <tag> [bench args]   (outputs under gpurun_out/sqk_<tag>/)
set -u
TAG=${1:-sqk}; shift || true
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sqk_$TAG
mkdir -p $OUT
CMD="python bench.py --path kernel --no-cpu-baseline --reps 2 $*"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  -f csv -d $OUT -o bench -- $CMD > $OUT/stdout.txt 2>&1
python - "$OUT" <<'PY'
import collections, csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    for nm in ("ik_quad_kernel", "ik_lane_kernel"):
        if nm in r["Kernel_Name"]:
            acc[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
last = acc[max(acc)]
wc = last["SQ_WAVE_CYCLES"]
print({k: v for k, v in last.items()})
print("per wave-cycle: VALU-busy %.3f  wait_any (parked at s_waitcnt / barrier) %.3f  wait_inst_any (issue stalls) %.3f" % (
    last["SQ_ACTIVE_INST_VALU"] / wc, last["SQ_WAIT_ANY"] / wc, last["SQ_WAIT_INST_ANY"] / wc))
print("instructions per wave: VALU %.0f  SALU %.0f  LDS %.0f" % (last["SQ_INSTS_VALU"] / last["SQ_WAVES"], last["SQ_INSTS_SALU"] / last["SQ_WAVES"], last["SQ_INSTS_LDS"] / last["SQ_WAVES"]))
PY

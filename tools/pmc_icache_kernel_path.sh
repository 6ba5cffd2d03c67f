#!/bin/bash
# Instruction-cache counters of the solve kernel of `bench.py --path kernel` (one PMC pass).
# Usage: bash tools/pmc_icache_kernel_path.sh <tag> [bench args]   (outputs under gpurun_out/ick_<tag>/)
set -u
TAG=${1:-ick}; shift || true
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ick_$TAG
mkdir -p $OUT
CMD="python bench.py --path kernel --no-cpu-baseline --reps 2 $*"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE \
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
print({k: v for k, v in last.items()})
req = max(last["SQC_ICACHE_REQ"], 1.0)
print("icache: hit rate %.4f  misses/req %.4f  dup-misses/req %.4f  ifetch per wave %.0f  mean fetches in flight / wave-cycle %.3f" % (
    last["SQC_ICACHE_HITS"] / req, last["SQC_ICACHE_MISSES"] / req, last["SQC_ICACHE_MISSES_DUPLICATE"] / req,
    last["SQ_IFETCH"] / max(last["SQ_WAVES"], 1.0), last["SQ_IFETCH_LEVEL"] / max(last["SQ_WAVE_CYCLES"], 1.0)))
PY

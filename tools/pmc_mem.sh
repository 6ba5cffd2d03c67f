#!/bin/bash
# Memory-pipeline counters per kernel (one PMC pass) for the bench run with one sub-pool.
# Usage: bash tools/pmc_mem.sh <tag>   (outputs under gpurun_out/mem_<tag>/)
set -u
TAG=${1:-mem}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/mem_$TAG
mkdir -p $OUT
export OPTIK_ENG_POOLS=1
CMD="python bench.py --steps 8 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_BUSY_avr GRBM_GUI_ACTIVE \
  -f csv -d $OUT -o bench -- $CMD > $OUT/stdout.txt 2>&1
python - "$OUT" <<'PY'
import collections, csv, glob, statistics, sys
f = glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
by = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    for nm in ("nnls", "eval", "update", "finish", "bucket"):
        if "eng_" + nm in r["Kernel_Name"]:
            k = (int(r["Dispatch_Id"]), nm)
            by[k][r["Counter_Name"]] = float(r["Counter_Value"])
            by[k]["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for nm in ("eval", "update", "bucket", "nnls", "finish"):
    ks = sorted((k for k in by if k[1] == nm), key=lambda k: -by[k]["dur_us"])[:100]
    if ks:
        print(nm, {c: round(statistics.median(by[k].get(c, 0) for k in ks), 1) for c in by[ks[0]]})
PY

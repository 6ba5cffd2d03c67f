#!/bin/bash
# Dynamic instruction mix of the dominant kernel at the driver's command: what the SQ counters can tell apart
# (f64 add / mul / fma / transcendental, int32, int64, conversions; SALU, LDS, SMEM, branches) -- each pass its own
# rocprofv3 run with --kernel-trace only.  The rest of the VALU stream (v_cndmask, v_mov, DPP moves, v_accvgpr_*) has no
# counter of its own: it is the residual, split by the static mix of tools/inst_mix.py.  Passes d / e: the instruction cache.
# Usage: bash tools/pmc_inst_mix.sh <tag> [bench args]   (outputs under gpurun_out/mix_<tag>/)
set -u
TAG=${1:-r6}; shift || true
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/mix_$TAG
mkdir -p $OUT/a $OUT/b $OUT/c $OUT/d $OUT/e
CMD="python bench.py --path kernel --no-cpu-baseline --no-other-configs --reps 3 $*"
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 \
  -f csv -d $OUT/a -o bench -- $CMD > $OUT/a/bench_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F32 \
  -f csv -d $OUT/b -o bench -- $CMD > $OUT/b/bench_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  -f csv -d $OUT/c -o bench -- $CMD > $OUT/c/bench_stdout.txt 2>&1
# (instruction cache: the kernel is 113 KB of code, the cache 64 KB per two CUs)
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE \
  -f csv -d $OUT/d -o bench -- $CMD > $OUT/d/bench_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU \
  -f csv -d $OUT/e -o bench -- $CMD > $OUT/e/bench_stdout.txt 2>&1
python - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
rec = {}
for d in "abcde":
    fs = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "ik_lane_kernel" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-3:]          # the timed repetitions' launches
    acc = collections.defaultdict(float)
    for r in rows:
        if int(r["Dispatch_Id"]) in ids:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    rec[d] = {k: v / max(len(ids), 1) for k, v in acc.items()}
json.dump(rec, open(f"{out}/inst_mix_counters.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
PY

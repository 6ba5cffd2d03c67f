#!/bin/bash
# rocprofv3 passes for the single-launch path of bench.py (ik_lane_kernel / ik_quad_kernel), as MI355X_MICROARCH.md
# prescribes (--pmc only ever with --kernel-trace; FETCH_SIZE and WRITE_SIZE in separate passes):
#   stats    kernel-trace --stats of the command
#   fetch / write / sq   one PMC pass each
# tools/pmc_kernel_path.py folds them into one record for the solve kernel's timed launches.
# Usage: bash tools/profile_kernel_path.sh <tag> [bench args]   (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r3k}; shift || true
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT/stats $OUT/fetch $OUT/write $OUT/sq $OUT/sq2
# (the repetition count is pinned: under the profiler a repetition takes longer, and pmc_kernel_path.py picks the timed
# launches as the LAST `reps` launches of the kernel in every pass)
CMD="python bench.py --path kernel --no-cpu-baseline --no-other-configs --reps 5 $*"
$CMD > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o bench -- $CMD > $OUT/stats/bench_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o bench -- $CMD > $OUT/fetch/bench_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/write -o bench -- $CMD > $OUT/write/bench_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES \
  -f csv -d $OUT/sq -o bench -- $CMD > $OUT/sq/bench_stdout.txt 2>&1
# (second SQ pass: lane activity of the VALU instructions -- SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) -- and where a wave waits)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU \
  -f csv -d $OUT/sq2 -o bench -- $CMD > $OUT/sq2/bench_stdout.txt 2>&1
python tools/pmc_kernel_path.py "$OUT" "$CMD"
head -6 $OUT/stats/bench_kernel_stats.csv | cut -c1-200

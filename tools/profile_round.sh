#!/bin/bash
# Round profile on the GPU box for ONE bench command: rocprofv3 kernel-trace stats, then separate
# PMC passes (FETCH_SIZE; WRITE_SIZE; SQ instruction counters) as MI355X_MICROARCH.md prescribes
# (--pmc only ever combined with --kernel-trace).  tools/pmc_by_command.py folds the passes
# into one record keyed by the bench's command key (bench.py:command_key), restricted to the
# launches of the TIMED run, so that bench.py can report PMC traffic for exactly the command
# it is running.
# Usage: bash tools/profile_round.sh <tag> [bench args]   (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r2}; shift || true
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT/stats $OUT/fetch $OUT/write $OUT/sq
CMD="python bench.py --no-cpu-baseline $*"
python bench.py --no-cpu-baseline "$@" > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o bench -- $CMD > $OUT/stats/bench_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o bench -- $CMD > $OUT/fetch/bench_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/write -o bench -- $CMD > $OUT/write/bench_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES \
  -f csv -d $OUT/sq -o bench -- $CMD > $OUT/sq/bench_stdout.txt 2>&1
python tools/pmc_by_command.py "$OUT" "$CMD"
head -9 $OUT/stats/bench_kernel_stats.csv | cut -c1-160

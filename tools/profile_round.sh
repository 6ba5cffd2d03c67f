#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel-trace stats of the default bench.py run,
# then two separate PMC passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes.
# Usage: bash tools/profile_round.sh <tag>   (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT/stats $OUT/fetch $OUT/write
CMD="python bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o bench -- $CMD > $OUT/stats/bench_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o bench -- $CMD > $OUT/fetch/bench_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/write -o bench -- $CMD > $OUT/write/bench_stdout.txt 2>&1
python tools/pmc_summary.py "$OUT" "$CMD"
head -8 $OUT/stats/bench_kernel_stats.csv | cut -c1-180

#!/bin/bash
# SQ instruction-mix counters per kernel (one PMC pass, 8 SQ slots) for the default bench run.
# Usage: bash tools/pmc_sq.sh <tag> [extra bench args]   (outputs under gpurun_out/sq_<tag>/)
set -u
TAG=${1:-sq}; shift || true
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
CMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  -f csv -d $OUT -o bench -- $CMD > $OUT/stdout.txt 2>&1
python tools/pmc_sq_summary.py $OUT > /dev/null; python tools/pmc_sq_top.py $OUT

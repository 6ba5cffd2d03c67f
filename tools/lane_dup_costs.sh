#!/bin/bash
# Cost of the evaluation, the NNLS phase, the BFGS update and the LSQ factor + bound rows of the lane-per-restart solver (ik_lane64.hpp) BY DUPLICATION: the
# variants run the phase twice on the same inputs (same results).  Build:
#   for v in EVAL NNLS BFGS LSQ; do python tools/build_lib_variant.py ldup_$v -DOPTIK_LANE_EXP_DUP_$v --only=ik_lane_kernel.o; done
k() { python bench.py --path kernel --no-cpu-baseline --steps 20 --warmup 5 --reps 3 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,3))'; }
base=$(k 2>/dev/null)
echo "product: $base M restarts/s"
for v in EVAL NNLS BFGS LSQ; do
  r=$(OPTIK_AMD_LIB=optik_amd/csrc/variants/ldup_$v.so k 2>/dev/null)
  python -c "b,r=$base,$r; print('ldup_$v: %.3f M  -> phase = %.1f %% of the product run time' % (r, 100.0*(b/r-1.0)))"
done

#!/bin/bash
# Cost of every phase of the quad solver's trip BY DUPLICATION: each optik_amd/csrc/variants/dup_<phase>.so runs
# that phase twice on the same inputs (ik_quad.hpp: OPTIK_QUAD_EXP_DUP_*, same results), so
# 1 / rate(variant) - 1 / rate(product) is the phase's time per restart at the driver's command.
# Build: for v in eval bfgs lsq nnls fin; do python tools/build_lib_variant.py dup_$v -DOPTIK_QUAD_EXP_DUP_${v^^} --only=ik_quad_throughput.o; done
k() { python bench.py --path kernel --no-cpu-baseline --steps 20 --warmup 5 --reps 3 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,3))'; }
base=$(k 2>/dev/null)
echo "product: $base M restarts/s"
for v in eval bfgs lsq nnls fin; do
  r=$(OPTIK_AMD_LIB=optik_amd/csrc/variants/dup_$v.so k 2>/dev/null)
  python -c "b,r=$base,$r; print('dup_$v: %.3f M  -> phase = %.1f %% of the product run time' % (r, 100.0*(b/r-1.0)))"
done

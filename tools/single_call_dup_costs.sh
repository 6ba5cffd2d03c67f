#!/bin/bash
# Share of every phase in the latency of a single optik_robot_ik call, BY DUPLICATION (as tools/quad_dup_costs.sh does
# for the throughput form): optik_amd/csrc/variants/latdup_<phase>.so runs that phase of the quad solver's latency
# object twice on the same inputs, same results.
# Build: for v in eval bfgs lsq nnls fin; do python tools/build_lib_variant.py latdup_$v -DOPTIK_QUAD_EXP_DUP_${v^^} --only=ik_quad_latency.o; done
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CALLS=${1:-1000}
lat() {  # $1 = library, $2 = parallelism ("" = unset)
  gcc -O2 -std=c11 -I"$ROOT/include" "$ROOT/tools/single_ik_latency.c" "$1" -Wl,-rpath,"$(dirname "$1")" -lm -o /tmp/lat_v
  /tmp/lat_v "$ROOT/optik_amd/robots/panda.urdf" panda_link0 panda_link8 "$CALLS" $2 | sed -n 's/Average time: \([0-9]*\)us.*/\1/p'
}
for par in "" 1; do
  base=$(lat "$ROOT/optik_amd/csrc/liboptik_amd.so" $par)
  echo "parallelism ${par:-unset}: product $base us"
  for v in eval bfgs lsq nnls fin; do
    r=$(lat "$ROOT/optik_amd/csrc/variants/latdup_$v.so" $par)
    python3 -c "b,r=$base,$r; print('  dup_$v: %d us -> phase = %.1f %% of the call' % (r, 100.0*(r-b)/b))"
  done
done

#!/bin/bash
# Latency of single optik_robot_ik calls through the C ABI (tools/single_ik_latency.c): Panda and UR10, first-success rule
# (parallelism unset) and deterministic rule (parallelism 1), with a GPU call between two ik calls and back to back.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CALLS=${1:-2000}
gcc -O2 -std=c11 -I"$ROOT/include" "$ROOT/tools/single_ik_latency.c" -L"$ROOT/optik_amd/csrc" -loptik_amd \
    -Wl,-rpath,"$ROOT/optik_amd/csrc" -lm -o /tmp/lat
for par in 0 1; do for gap in 1 0; do
  /tmp/lat "$ROOT/optik_amd/robots/panda.urdf" panda_link0 panda_link8 "$CALLS" $par $gap
done; done
for par in 0 1; do /tmp/lat "$ROOT/optik_amd/robots/ur10.urdf" base_link ee_link "$CALLS" $par 1; done

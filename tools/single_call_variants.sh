#!/bin/bash
# Single optik_robot_ik latency of library variants side by side on one box (tools/build_lib_variant.py builds them).
# Usage: tools/single_call_variants.sh <calls> <variant> [<variant> ...]   (the product library is always measured first)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CALLS=${1:-2000}; shift
lat() { gcc -O2 -std=c11 -I"$ROOT/include" "$ROOT/tools/single_ik_latency.c" "$1" -Wl,-rpath,"$(dirname "$1")" -lm -o /tmp/lat_v
        /tmp/lat_v "$ROOT/optik_amd/robots/panda.urdf" panda_link0 panda_link8 "$CALLS" $2 | sed -n 's/Average time: \([0-9]*\)us.*/\1/p'; }
for rep in 1 2; do
  for par in "" 1; do
    line="parallelism ${par:-unset}: product $(lat "$ROOT/optik_amd/csrc/liboptik_amd.so" $par)"
    for v in "$@"; do line="$line  $v $(lat "$ROOT/optik_amd/csrc/variants/$v.so" $par)"; done
    echo "$line us"
  done
done

#!/bin/bash
# Builds tuning variants of liboptik_amd.so into optik_amd/csrc/variants/<name>.so.
# Usage: tools/build_variants.sh name1:"-DA=1 -DB=2" name2:"..." ...
cd "$(dirname "$0")/../optik_amd/csrc" || exit 1
mkdir -p variants
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -pthread \
      $defs -x hip ik_kernels.hip robot_host.cpp -o variants/$name.so 2>variants/$name.log && echo "built $name" || echo "FAILED $name" ) &
done
wait

#!/bin/bash
# Builds optik_amd/csrc/variants/<name>.so: the library with ik_quad_kernel.hip recompiled with extra
# defines (seconds), linked against the current objects of the other translation units.
# Usage: tools/build_quad_variant.sh name "-DOPTIK_QUAD_WAVES=2 ..."   (run optik_amd/build.py first)
cd "$(dirname "$0")/../optik_amd/csrc" || exit 1
mkdir -p variants
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -pthread $* \
  -Rpass-analysis=kernel-resource-usage -x hip -c ik_quad_kernel.hip -o variants/$name.quad.o 2> variants/$name.log || { echo "FAILED $name"; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -pthread ik_kernels.o variants/$name.quad.o robot_host.o -o variants/$name.so
grep -A10 "ik_quad_kernelILi7ELb1ELi2" variants/$name.log | grep "VGPRs:\|AGPRs:\|Scratch\|Occupancy" | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - -

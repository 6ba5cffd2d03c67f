#!/bin/bash
# A / B / A / B of the default bench command between the product library and one variant (tools/build_lib_variant.py),
# on one box.  Usage: tools/ab_kernel_path.sh <variant> [bench args]
V=$1; shift
k() { python bench.py --no-cpu-baseline --steps 20 --warmup 5 --reps 3 "$@" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,3))'; }
for r in 1 2 3; do
  echo "product $(k "$@" 2>/dev/null)   $V $(OPTIK_AMD_LIB=optik_amd/csrc/variants/$V.so k "$@" 2>/dev/null)"
done

#!/bin/bash
# The default bench command on the product library and on variants (tools/build_lib_variant.py), interleaved three
# times on one box, the order ROTATED from row to row (the first run of a row measured ~0.3 % low in round 5: whatever
# runs first pays for it once).  Usage: tools/ab_kernel_path.sh "<variant> [<variant> ...]" [bench args]
names=(product $1); shift
k() { python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --reps 9 "$@" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,3))'; }
n=${#names[@]}
for r in 0 1 2; do
  declare -A res
  for i in $(seq 0 $((n-1))); do
    V=${names[$(( (i + r) % n ))]}
    if [ "$V" = product ]; then res[$V]=$(k "$@" 2>/dev/null); else res[$V]=$(OPTIK_AMD_LIB=optik_amd/csrc/variants/$V.so k "$@" 2>/dev/null); fi
  done
  line=""; for V in "${names[@]}"; do line="$line$V ${res[$V]}   "; done
  echo "$line"
done

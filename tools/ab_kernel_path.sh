#!/bin/bash
# The default bench command on the product library and on variants (tools/build_lib_variant.py), interleaved three
# times on one box.  Usage: tools/ab_kernel_path.sh "<variant> [<variant> ...]" [bench args]
VS=$1; shift
k() { python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --reps 9 "$@" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,3))'; }
for r in 1 2 3; do
  line="product $(k "$@" 2>/dev/null)"
  for V in $VS; do line="$line   $V $(OPTIK_AMD_LIB=optik_amd/csrc/variants/$V.so k "$@" 2>/dev/null)"; done
  echo "$line"
done

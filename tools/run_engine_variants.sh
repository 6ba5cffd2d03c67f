#!/bin/bash
# engine path with ik_kernels.hip built under other scheduling strategies (optik_amd/csrc/variants/eng_*.so), same box
one() { python bench.py --no-cpu-baseline --steps 20 --warmup 5 --reps 3 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,2), [round(v/1e6,2) for v in d["config"]["value_reps"]])'; }
echo "product: $(one 2>/dev/null)"
for v in optik_amd/csrc/variants/eng_*.so; do echo "$v: $(OPTIK_AMD_LIB=$v one 2>/dev/null)"; done
echo "product: $(one 2>/dev/null)"

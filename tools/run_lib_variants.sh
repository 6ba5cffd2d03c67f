#!/bin/bash
# kernel path, engine path and single-call latency of the product library and of every optik_amd/csrc/variants/*.so, same box
k() { python bench.py --path kernel --no-cpu-baseline --steps 20 --warmup 5 --reps 3 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,2))'; }
k48() { python bench.py --path kernel --no-cpu-baseline --steps 48 --warmup 5 --reps 3 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,2))'; }
e() { python bench.py --no-cpu-baseline --steps 20 --warmup 5 --reps 3 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,2))'; }
l() { python tools/single_ik_latency.py 600 1 | grep -o "Average time: [0-9]*us"; }
echo "product: kernel $(k 2>/dev/null) / $(k48 2>/dev/null)  engine $(e 2>/dev/null)  $(l 2>/dev/null)"
for v in optik_amd/csrc/variants/*.so; do export OPTIK_AMD_LIB=$v; echo "$v: kernel $(k 2>/dev/null) / $(k48 2>/dev/null)  engine $(e 2>/dev/null)  $(l 2>/dev/null)"; done
unset OPTIK_AMD_LIB
echo "product: kernel $(k 2>/dev/null) / $(k48 2>/dev/null)  engine $(e 2>/dev/null)  $(l 2>/dev/null)"

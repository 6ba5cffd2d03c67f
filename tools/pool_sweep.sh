#!/bin/bash
# engine pool size / sub-pool count / hand-over threshold against the 20-step bench line (same box)
one() { python bench.py --no-cpu-baseline --steps 20 --warmup 5 --reps 3 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,2), [round(v/1e6,2) for v in d["config"]["value_reps"]])'; }
echo "default: $(one 2>/dev/null)"
for s in 196608 262144 327680 458752; do echo "slots=$s: $(OPTIK_ENGINE_SLOTS=$s one 2>/dev/null)"; done
for p in 2 4; do echo "pools=$p: $(OPTIK_ENG_POOLS=$p one 2>/dev/null)"; done
echo "slots=262144 tail_max=65536: $(OPTIK_ENGINE_SLOTS=262144 OPTIK_ENG_TAIL_MAX=65536 one 2>/dev/null)"
echo "depth=3: $(OPTIK_ENG_DEPTH=3 one 2>/dev/null)"
echo "default again: $(one 2>/dev/null)"

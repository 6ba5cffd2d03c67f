one() { python bench.py --no-cpu-baseline --steps 20 --warmup 5 --reps 3 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,2), [round(v/1e6,2) for v in d["config"]["value_reps"]])'; }
echo "default: $(one 2>/dev/null)"
for w in 4 5 6; do echo "tail waves/CU=$w: $(OPTIK_ENG_TAIL_WAVES=$w one 2>/dev/null)"; done
echo "tail waves/CU=4 tail_max=65536: $(OPTIK_ENG_TAIL_WAVES=4 OPTIK_ENG_TAIL_MAX=65536 one 2>/dev/null)"
echo "tail waves/CU=4 tail_max=32768: $(OPTIK_ENG_TAIL_WAVES=4 OPTIK_ENG_TAIL_MAX=32768 one 2>/dev/null)"
echo "default: $(one 2>/dev/null)"

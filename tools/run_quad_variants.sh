k() { python bench.py --path kernel --no-cpu-baseline --steps 20 --warmup 5 --reps 3 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,2), [round(v/1e6,2) for v in d["config"]["value_reps"]])'; }
echo "product: $(k 2>/dev/null)"
for v in optik_amd/csrc/variants/*.so; do echo "$v: $(OPTIK_AMD_LIB=$v k 2>/dev/null)"; done

one() { python bench.py --path kernel --no-cpu-baseline --steps 20 --warmup 5 --reps 3 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,2), [round(v/1e6,2) for v in d["config"]["value_reps"]])'; }
echo "product: $(one 2>/dev/null)"
for v in base ilp itmin itilp nopost memcl; do echo "$v: $(OPTIK_AMD_LIB=optik_amd/csrc/variants/$v.so one 2>/dev/null)"; done

python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine_fuzz.py -x -q 2>&1 | tail -3
for tm in 32768 65536 98304 131072 196608 262144; do
  echo "quad tail_max=$tm: $(OPTIK_ENG_TAIL_MAX=$tm python bench.py --no-cpu-baseline --steps 20 --warmup 5 --reps 3 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,2), [round(v/1e6,2) for v in d["config"]["value_reps"]])')"
done
echo "coop tail: $(OPTIK_ENG_TAIL=coop python bench.py --no-cpu-baseline --steps 20 --warmup 5 --reps 3 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e6,2), [round(v/1e6,2) for v in d["config"]["value_reps"]])')"

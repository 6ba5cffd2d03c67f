#!/bin/bash
# Round 5: the spill threshold of the lane-per-restart form (ik_spill.hpp, option spill_at): bench line + the isolated
# config-2 launch per setting.  Usage: bash tools/spill_sweep.sh "0 16 32 48" [extra bench args]
set -u
VALS=${1:-"0 16 32 48"}; shift || true
for v in $VALS; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --set-option spill_at=$v "$@" > /tmp/sp_$v.json 2>/tmp/sp_$v.err || { tail -5 /tmp/sp_$v.err; continue; }
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f"/tmp/sp_{v}.json").read().strip().splitlines()[-1])
oc = d["config"]["other_configs"] or {}
g = lambda *k: (lambda x: x)(__import__("functools").reduce(lambda a, b: (a or {}).get(b) if isinstance(a, dict) else None, k, oc))
print(f"spill_at={v:>3}: {d['value']/1e6:7.3f} M restarts/s  kernel_ms {d['roofline']['kernel_ms']:.3f}  single launch {g('config2_single_launch','single_launch_ms') or 0:.3f} ms  "
      f"ur10 {(g('config3_ur10_1M','tol_f_1e-6','restarts_per_s') or 0)/1e6:.2f} M  cfg4 shard {(g('config4_one_gpu_shard','restarts_per_s') or 0)/1e6:.2f} M  "
      f"cfg5 4096 det/any {(g('config5_all_4096_targets','ik_calls_per_s_deterministic') or 0)/1e6:.3f}/{(g('config5_all_4096_targets','ik_calls_per_s_find_any') or 0)/1e6:.3f} M", flush=True)
PY
done

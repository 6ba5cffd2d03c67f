#!/bin/bash
# spill_at x spill_quads (ik_spill.hpp): "at:quads" pairs
set -u
for pr in ${1:-"0:16 2:2 4:4 8:4 8:8 16:8"}; do
  at=${pr%%:*}; q=${pr##*:}
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --set-option spill_at=$at --set-option spill_quads=$q > /tmp/sp.json 2>/tmp/sp.err || { tail -5 /tmp/sp.err; continue; }
  python - "$at" "$q" <<'PY'
import json, sys
d = json.loads(open("/tmp/sp.json").read().strip().splitlines()[-1])
oc = d["config"]["other_configs"]
print(f"spill_at={sys.argv[1]:>3} quads={sys.argv[2]:>2}: {d['value']/1e6:7.3f} M restarts/s  kernel_ms {d['roofline']['kernel_ms']:.3f}  single launch {oc['config2_single_launch']['single_launch_ms']:.3f} ms  "
      f"ur10 {oc['config3_ur10_1M']['tol_f_1e-6']['restarts_per_s']/1e6:.2f} M  cfg4 shard {oc['config4_one_gpu_shard']['restarts_per_s']/1e6:.2f} M", flush=True)
PY
done

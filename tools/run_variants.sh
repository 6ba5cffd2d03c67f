#!/bin/bash
# On the GPU box: runs bench.py once per variant library (copied over the in-tree library
# of this scratch checkout).  Usage: tools/run_variants.sh "<bench args>" name1 name2 ...
ARGS=$1; shift
cp optik_amd/csrc/liboptik_amd.so /tmp/liboptik_amd_base.so
for v in "$@"; do
  cp optik_amd/csrc/variants/$v.so optik_amd/csrc/liboptik_amd.so
  python bench.py --no-cpu-baseline $ARGS 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', round(d['value']/1e6,3),'M/s', 'ms/step',round(d['ms_per_step'],3), r.get('all_kernels_ms'), 'trips', r.get('trips'))"
done
cp /tmp/liboptik_amd_base.so optik_amd/csrc/liboptik_amd.so

#!/bin/bash
# A/B of two builds of the library on the same box: $1 = alternative .so (relative to the repo)
for lib in ratatosk_amd/libratatosk_hip.so $1; do
  cp $lib /tmp/cur.so
  for i in 1 2; do RTK_LIB_OVERRIDE=$PWD/$lib timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs --serial 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['roofline']['kernel_ms_per_step']['k_regions'])"; done
done

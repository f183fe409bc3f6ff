#!/bin/bash
# A/B of environment knobs on the default build (configs[1], serial): each argument is "VAR=value"
for kv in "X=0" "$@"; do
  for i in 1 2; do env $kv RTK_TRACE=1 timeout 300 python bench.py --config1-only --steps 4 --warmup 1 --no-cpu-baseline --no-host-legs --serial 2>gpurun_out/env_trace.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$kv]', round(d['value']/1e9,4), d['roofline']['kernel_ms_per_step']['k_regions'], d['roofline']['regions_redone_bigger_arena'])"; done
  grep "k_regions attempt\|work area" gpurun_out/env_trace.txt | tail -2
done

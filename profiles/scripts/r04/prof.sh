#!/bin/bash
# lap profile of k_regions (PROF variant) + A/B of variants given as arguments (names under ratatosk_amd/variants)
mkdir -p gpurun_out/r04_prof
RTK_LIB_OVERRIDE=$PWD/ratatosk_amd/variants/libratatosk_hip_prof.so RTK_TRACE=1 timeout 600 python bench.py --config1-only --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs --serial 2> gpurun_out/r04_prof/trace.txt | tail -c 600
grep "lap profile\|k_regions attempt" gpurun_out/r04_prof/trace.txt | tail -2
for v in "" "$@"; do
  lib=ratatosk_amd/libratatosk_hip.so; [ -n "$v" ] && lib=ratatosk_amd/variants/libratatosk_hip_$v.so
  for i in 1 2; do RTK_LIB_OVERRIDE=$PWD/$lib timeout 300 python bench.py --config1-only --steps 4 --warmup 1 --no-cpu-baseline --no-host-legs --serial 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v]', round(d['value']/1e9,4), d['roofline']['kernel_ms_per_step'])"; done
done

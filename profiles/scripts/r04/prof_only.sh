#!/bin/bash
for w in "" $@; do
echo "== RTK_REGION_WAVES=$w"
RTK_REGION_WAVES=$w RTK_LIB_OVERRIDE=$PWD/ratatosk_amd/variants/libratatosk_hip_prof.so RTK_TRACE=1 timeout 600 python bench.py --config1-only --steps 2 --warmup 1 --no-cpu-baseline --no-host-legs --serial 2>&1 >/dev/null | grep "in-situ\|k_regions attempt" | tail -2
done

#!/bin/bash
# A/B on one box: Myers golden tests once, then serial bench lines of the default build and of the variants named as arguments
timeout 600 python -m pytest tests/test_gpu_myers.py tests/test_myers_band.py -m gpu -x -q 2>&1 | tail -2
for v in "" "$@"; do
  lib=ratatosk_amd/libratatosk_hip.so; [ -n "$v" ] && lib=ratatosk_amd/variants/libratatosk_hip_$v.so
  for i in 1 2; do RTK_LIB_OVERRIDE=$PWD/$lib timeout 300 python bench.py --config1-only --steps 4 --warmup 1 --no-cpu-baseline --no-host-legs --serial 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v]', round(d['value']/1e9,4), d['roofline']['kernel_ms_per_step']['k_regions'], d['roofline']['k_regions_wave_cycle_share'])"; done
done

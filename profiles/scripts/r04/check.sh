#!/bin/bash
# parity (correction + seeds + configs) then the serial configs[1] line with kernel times; arguments = variants to A/B
timeout 1200 python -m pytest tests/test_gpu_correct.py tests/test_configs.py tests/test_toy_golden.py -m gpu -x -q 2>&1 | tail -3
for v in "" "$@"; do
  lib=ratatosk_amd/libratatosk_hip.so; [ -n "$v" ] && lib=ratatosk_amd/variants/libratatosk_hip_$v.so
  for i in 1 2; do RTK_LIB_OVERRIDE=$PWD/$lib timeout 300 python bench.py --config1-only --steps 4 --warmup 1 --no-cpu-baseline --no-host-legs --serial 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v]', round(d['value']/1e9,4), round(d['ms_per_step'],2), d['roofline']['kernel_ms_per_step'])"; done
done

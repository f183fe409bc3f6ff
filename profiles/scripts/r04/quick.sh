#!/bin/bash
# quick GPU check: parity tests of the correction path, then serial bench lines of the default build and of the variants named as arguments
timeout 900 python -m pytest tests/test_gpu_correct.py tests/test_gpu_myers.py tests/test_configs.py -m gpu -x -q 2>&1 | tail -3
for v in "" "$@"; do
  lib=ratatosk_amd/libratatosk_hip.so; [ -n "$v" ] && lib=ratatosk_amd/variants/libratatosk_hip_$v.so
  for i in 1 2; do RTK_LIB_OVERRIDE=$PWD/$lib RTK_TRACE=1 timeout 300 python bench.py --config1-only --steps 4 --warmup 1 --no-cpu-baseline --no-host-legs --serial 2>gpurun_out/trace_$v.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v]', round(d['value']/1e9,4), d['roofline']['kernel_ms_per_step'])"; done
  grep "lap profile" gpurun_out/trace_$v.txt | tail -1
done

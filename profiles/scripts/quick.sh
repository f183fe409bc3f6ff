#!/bin/bash
# quick GPU check: parity tests of the correction path + short bench lines (kernel times, k_regions cycle shares)
timeout 600 python -m pytest tests/test_gpu_correct.py tests/test_gpu_myers.py -m gpu -x -q 2>&1 | tail -2
for mode in "" "--serial"; do
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $mode 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$mode', d['value'], d['ms_per_step']); r=d['roofline']; print(r['kernel_ms_per_step']); print(r['k_regions_wave_cycle_share'])"
done

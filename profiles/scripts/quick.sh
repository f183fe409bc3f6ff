#!/bin/bash
# quick GPU check: parity tests of the correction path + a short bench line (kernel times, k_regions cycle shares)
timeout 600 python -m pytest tests/test_gpu_correct.py tests/test_gpu_myers.py -m gpu -x -q 2>&1 | tail -2
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); r=d['roofline']; print(r['kernel_ms_per_step']); print(r['k_regions_wave_cycle_share'])"

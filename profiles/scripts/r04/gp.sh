#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_myers.py tests/test_myers_band.py tests/test_gpu_correct.py tests/test_gpu_seeds.py tests/test_gpu_lookup.py -m gpu -x -q 2>&1 | tail -5
bash profiles/scripts/r04_ab_nocheck.sh noglob

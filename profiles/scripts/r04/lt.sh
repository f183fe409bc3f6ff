#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_myers.py tests/test_myers_band.py tests/test_gpu_correct.py -m gpu -x -q 2>&1 | tail -15
bash profiles/scripts/r04_ab_nocheck.sh nolt

#!/bin/bash
# new GPU tests of round 4 (index vs independent oracle, [D1] / [A2] / [A3] switches, the [A3] toy), then the readings counts
mkdir -p gpurun_out/r04_parity
timeout 1500 python -m pytest tests/test_index_build.py tests/test_d1_switch.py tests/test_a2_switch.py tests/test_a3_switch.py tests/test_toy_golden.py -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r04_parity/tests.log
timeout 1200 python profiles/scripts/r04_readings_count.py gpurun_out/r04_parity > gpurun_out/r04_parity/counts.json 2> gpurun_out/r04_parity/counts.err; tail -c 2500 gpurun_out/r04_parity/counts.json; tail -3 gpurun_out/r04_parity/counts.err

#!/bin/bash
# round-4 baseline: bench line (serial) with RTK_TRACE, per-class cycle shares of k_regions
mkdir -p gpurun_out/r04_base
RTK_TRACE=1 timeout 900 python bench.py --config1-only --steps 4 --warmup 1 --no-cpu-baseline --no-host-legs --serial > gpurun_out/r04_base/bench_serial.json 2> gpurun_out/r04_base/trace_serial.txt
tail -c 3000 gpurun_out/r04_base/bench_serial.json
grep "rtk trace" gpurun_out/r04_base/trace_serial.txt | tail -12
bash profiles/scripts/trace_classes.sh > gpurun_out/r04_base/classes.txt 2>&1
cat gpurun_out/r04_base/classes.txt

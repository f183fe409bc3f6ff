#!/bin/bash
# Developer run: the configs[1] line of the bench with its per-kernel times (one step at a time), and the same with RTK_TRACE laps of one step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --config1-only --no-cpu-baseline --no-host-legs --steps 10 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.4g ms_per_step %.2f' % (d['value'], d['ms_per_step'])); print(d['roofline']['kernel_ms_per_step'])" > gpurun_out/r05_bench_quick.txt 2>&1
RTK_TRACE=1 python bench.py --config1-only --no-cpu-baseline --no-host-legs --steps 3 --warmup 1 2>&1 | grep -E "regions (alloc|enum|regions|done)|seeds attempt" | tail -12 >> gpurun_out/r05_bench_quick.txt
cat gpurun_out/r05_bench_quick.txt

#!/bin/bash
# round 3: reads that depend on the reading of [A2]; N = 1 on the configs[2] graph; size-class trace of k_regions (developer)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python profiles/scripts/a2_count.py > gpurun_out/a2_count.json 2> gpurun_out/a2_count.err; tail -2 gpurun_out/a2_count.err; cat gpurun_out/a2_count.json
timeout 900 python bench.py --config2 --no-cpu-baseline --no-host-legs > gpurun_out/bench_config2.json 2> gpurun_out/bench_config2.err; tail -2 gpurun_out/bench_config2.err; cut -c1-400 gpurun_out/bench_config2.json
RTK_TRACE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-legs --serial 2>&1 >/dev/null | grep "size class\|fine shares\|DFS book\|shares of\|k_regions attempt" | tail -6 > gpurun_out/trace_classes.txt; cat gpurun_out/trace_classes.txt

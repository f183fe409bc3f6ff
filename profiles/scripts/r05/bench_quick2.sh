#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( RTK_TRACE=1 python bench.py --config1-only --no-cpu-baseline --no-host-legs --steps 3 --warmup 1 2>&1 | grep -E "k_regions attempt|region work area|k_regions waves" | tail -6
  echo "== A/B script"
  RTK_TRACE=1 python profiles/scripts/r05_lanes_ab.py c1 64000000 0 2>&1 | grep -E "k_regions attempt|region work area|k_regions waves|gap<" | tail -6 ) > gpurun_out/r05_bench_quick2.txt 2>&1
cat gpurun_out/r05_bench_quick2.txt

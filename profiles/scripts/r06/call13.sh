#!/bin/bash
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash profiles/scripts/r06/pass2_trace.sh > /dev/null 2>&1; cp gpurun_out/r06_pass2_trace.txt gpurun_out/r06_pass2_trace_final.txt; grep -E "== run|new device" gpurun_out/r06_pass2_trace_final.txt
( time timeout 2400 python bench.py --workdir /tmp/rtk_r06_wd > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err ) 2> gpurun_out/r06_bench_time.txt
tail -2 gpurun_out/r06_bench.err; cat gpurun_out/r06_bench_time.txt

#!/bin/bash
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_cli.py tests/test_pass2.py tests/test_coalesce.py tests/test_configs.py -q -m gpu 2>&1 | tail -3 > gpurun_out/r06_last_tests.txt; cat gpurun_out/r06_last_tests.txt
bash profiles/scripts/r06/pass2_trace.sh > /dev/null 2>&1; cp gpurun_out/r06_pass2_trace.txt gpurun_out/r06_pass2_trace_reserve11.txt; grep -E "== run|new device" gpurun_out/r06_pass2_trace_reserve11.txt
( time timeout 2400 python bench.py --workdir /tmp/rtk_wd > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err ) 2> gpurun_out/r06_bench_time.txt
tail -2 gpurun_out/r06_bench.err; cat gpurun_out/r06_bench_time.txt

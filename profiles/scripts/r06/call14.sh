#!/bin/bash
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash profiles/scripts/r06/sdma_ab.sh
( time timeout 2400 python bench.py --workdir /tmp/rtk_wd > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err ) 2> gpurun_out/r06_bench_time.txt
tail -2 gpurun_out/r06_bench.err; cat gpurun_out/r06_bench_time.txt

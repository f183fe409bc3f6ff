#!/bin/bash
# the bench line alone, as the driver runs it (after the profile set: it quotes profiles/r05_pmc_summary.json)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles_r05
( time timeout 1500 python bench.py > gpurun_out/profiles_r05/r05_bench.json 2> gpurun_out/profiles_r05/r05_bench.err ) 2> gpurun_out/profiles_r05/r05_bench_time.txt
tail -3 gpurun_out/profiles_r05/r05_bench_time.txt; tail -c 400 gpurun_out/profiles_r05/r05_bench.json

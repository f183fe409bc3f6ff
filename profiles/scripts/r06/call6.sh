#!/bin/bash
# Round 6, GPU call 6: coalescing with the bounded wait in every case and the cold-start split; then configs[4] through bench_config4.py (3 Gb, 16 distinct tickets)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/rtk_wd; O=gpurun_out/r06_call6.txt; : > $O
timeout 600 python -m pytest tests/test_coalesce.py -x -q -m gpu 2>&1 | tail -2 >> $O
timeout 1500 python profiles/scripts/r06/ticket_sizes.py /tmp/rtk_wd > gpurun_out/r06_ticket_sizes.txt 2> gpurun_out/r06_ticket_sizes.err; tail -4 gpurun_out/r06_ticket_sizes.txt >> $O
( time timeout 1500 python bench_config4.py gpurun_out/r06_config4.json 3000 30 16 128 /tmp/rtk_c4 > gpurun_out/r06_config4.log 2>&1 ) 2>> $O
tail -c 1500 gpurun_out/r06_config4.log >> $O
cat $O

#!/bin/bash
# D2H copies through the SDMA engines (default) or through copy kernels (HSA_ENABLE_SDMA=0): small tickets through rtk_correct_batch and the second pass (14-19 of 148 fetches of a
# second-pass run take > 5 ms on the default route)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/rtk_wd; O=gpurun_out/r06_sdma_ab.txt; : > $O
for rep in 1 2; do
  timeout 500 python profiles/scripts/r06/tickets_quick.py /tmp/rtk_wd "sdma (default)" 2>/dev/null >> $O
  HSA_ENABLE_SDMA=0 timeout 500 python profiles/scripts/r06/tickets_quick.py /tmp/rtk_wd "HSA_ENABLE_SDMA=0" 2>/dev/null >> $O
done
bash profiles/scripts/r06/pass2_trace.sh > /dev/null 2>&1; grep -E "== run|slow fetches" gpurun_out/r06_pass2_trace.txt | sed 's/^/sdma (default) /' >> $O
HSA_ENABLE_SDMA=0 bash profiles/scripts/r06/pass2_trace.sh > /dev/null 2>&1; grep -E "== run|slow fetches" gpurun_out/r06_pass2_trace.txt | sed 's/^/HSA_ENABLE_SDMA=0 /' >> $O
cat $O

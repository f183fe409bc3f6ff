#!/bin/bash
# Round 6, GPU call 3: parity of the new k_inexact (keys / hit staging off the stack) and of the coalescing entry, its times, bases/s by ticket size with the
# stage-aware gathering, and the CU-partition probe (seed stage of step s+1 on its own CUs beside the persistent region kernel of step s).
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=gpurun_out/r06_call3.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_seeds.py tests/test_gpu_correct.py tests/test_capi.py tests/test_coalesce.py -x -q -m gpu 2>&1 | tail -5 >> $O
W=/tmp/rtk_wd; mkdir -p $W
B="python bench.py --workdir $W --steps 6 --warmup 2 --no-cpu-baseline --no-host-legs --no-config1-leg"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 ms_per_step %.2f value %.4g' % (d['ms_per_step'], d['value']), d['roofline']['kernel_ms_per_step'])"; }
timeout 900 $B --serial 2>/dev/null | line "serial default" >> $O
timeout 1500 python profiles/scripts/r06/ticket_sizes.py $W > gpurun_out/r06_ticket_sizes.txt 2> gpurun_out/r06_ticket_sizes.err; tail -4 gpurun_out/r06_ticket_sizes.txt >> $O
P=gpurun_out/r06_cu_mask_probe.txt; : > $P
timeout 600 $B --serial 2>/dev/null | line "serial, no partition" >> $P
timeout 600 $B --overlap 2>/dev/null | line "overlap, no partition" >> $P
for cfg in "16 3840" "24 3712" "32 3584" "48 3328" "24,i 3712" "32,i 3584"; do set -- $cfg
  RTK_CU_SPLIT=$1 RTK_REGION_WAVES=$2 timeout 600 $B --overlap 2>/dev/null | line "overlap, RTK_CU_SPLIT=$1 RTK_REGION_WAVES=$2" >> $P
done
RTK_CU_SPLIT=24 RTK_REGION_WAVES=3712 timeout 600 $B --serial 2>/dev/null | line "serial, RTK_CU_SPLIT=24 RTK_REGION_WAVES=3712" >> $P
cat $O $P

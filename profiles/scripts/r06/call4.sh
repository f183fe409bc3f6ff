#!/bin/bash
# Round 6, GPU call 4: two region stages side by side on half of the work areas each (small tickets), paced coalescing; then the occupancy variants of k_regions
# with the path search as calls (fewer spills at 80 / 72 registers?)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=gpurun_out/r06_call4.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_correct.py tests/test_capi.py tests/test_coalesce.py -x -q -m gpu 2>&1 | tail -5 >> $O
W=/tmp/rtk_wd; mkdir -p $W
B="python bench.py --workdir $W --steps 6 --warmup 2 --no-cpu-baseline --no-host-legs --no-config1-leg"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 ms_per_step %.2f value %.4g' % (d['ms_per_step'], d['value']), d['roofline']['kernel_ms_per_step'])"; }
timeout 900 $B --serial 2>/dev/null | line "serial default" >> $O
timeout 1500 python profiles/scripts/r06/ticket_sizes.py $W > gpurun_out/r06_ticket_sizes.txt 2> gpurun_out/r06_ticket_sizes.err; tail -4 gpurun_out/r06_ticket_sizes.txt >> $O
RTK_HALF_SLAB_BASES=0 timeout 1500 python profiles/scripts/r06/ticket_sizes.py $W 2>/dev/null | tail -4 | sed 's/^/no halves: /' >> $O
P=gpurun_out/r06_wpe_probe2.txt; : > $P
for cfg in "wpe6 6144" "wpe6s 6144" "wpe6sh 6144" "wpe7 7168" "wpe7s 7168"; do set -- $cfg
  RTK_LIB_OVERRIDE=$PWD/ratatosk_amd/variants/libratatosk_hip_$1.so RTK_REGION_WAVES=$2 timeout 600 $B --serial 2>/dev/null | line "$1 waves=$2" >> $P
done
cat $O $P

#!/bin/bash
# Round 6, GPU call 2: the new gpu tests (rtk_correct_batch through ctypes, ticket coalescing), bases/s by ticket size through rtk_correct_batch,
# then the occupancy probe again (core dumps off: a faulting variant filled the box's disk in call 1).
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=gpurun_out/r06_call2.txt; : > $O
timeout 900 python -m pytest tests/test_capi.py tests/test_coalesce.py -x -q -m gpu 2>&1 | tail -5 >> $O
W=/tmp/rtk_wd; mkdir -p $W
timeout 1500 python profiles/scripts/r06/ticket_sizes.py $W > gpurun_out/r06_ticket_sizes.txt 2> gpurun_out/r06_ticket_sizes.err; tail -5 gpurun_out/r06_ticket_sizes.txt >> $O; tail -3 gpurun_out/r06_ticket_sizes.err >> $O
P=gpurun_out/r06_wpe_probe.txt; : > $P
run() { # name, lib, waves
  ( [ -n "$2" ] && export RTK_LIB_OVERRIDE=$PWD/ratatosk_amd/variants/$2; [ -n "$3" ] && export RTK_REGION_WAVES=$3
    timeout 600 python bench.py --workdir $W --steps 6 --warmup 2 --no-cpu-baseline --no-host-legs --no-config1-leg --serial 2> gpurun_out/r06_wpe_$1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 waves=${3:-4096} ms_per_step %.2f value %.4g' % (d['ms_per_step'], d['value']), d['roofline']['kernel_ms_per_step'])" ) >> $P 2>&1
  tail -2 gpurun_out/r06_wpe_$1.err >> $P
}
for v in wpe5 wpe6; do RTK_LIB_OVERRIDE=$PWD/ratatosk_amd/variants/libratatosk_hip_$v.so RTK_REGION_WAVES=$([ $v = wpe5 ] && echo 5120 || echo 6144) timeout 600 python -m pytest tests/test_gpu_correct.py -x -q -k "volume or branching" 2>&1 | tail -3 | sed "s/^/$v parity: /" >> $P; done
run default "" ""
run wpe5 libratatosk_hip_wpe5.so 5120
run wpe6 libratatosk_hip_wpe6.so 6144
run wpe5_4096 libratatosk_hip_wpe5.so 4096
df -h /tmp | tail -1 >> $P
cat $O $P

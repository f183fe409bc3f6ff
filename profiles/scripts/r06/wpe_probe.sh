#!/bin/bash
# Round 6 probe: does k_regions pay for more waves per SIMD today? default build (128 VGPRs, 4 waves/SIMD, 4096 waves) against the whole-TU builds of
# profiles/scripts/build_wpe_variant.sh at 5 (96 VGPRs, 5120 waves) and 6 (80 VGPRs, 6144 waves) waves per SIMD; 60 Mb set, one step at a time.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; O=gpurun_out/r06_wpe_probe.txt; : > $O
W=/tmp/rtk_wd; mkdir -p $W
run() { # name, lib, waves
  ( [ -n "$2" ] && export RTK_LIB_OVERRIDE=$PWD/ratatosk_amd/variants/$2; [ -n "$3" ] && export RTK_REGION_WAVES=$3
    timeout 900 python bench.py --workdir $W --steps 6 --warmup 2 --no-cpu-baseline --no-host-legs --no-config1-leg --serial 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 waves=${3:-4096} ms_per_step %.2f value %.4g' % (d['ms_per_step'], d['value']), d['roofline']['kernel_ms_per_step'])" ) >> $O 2>&1
}
run default "" ""
run wpe5 libratatosk_hip_wpe5.so 5120
run wpe6 libratatosk_hip_wpe6.so 6144
run wpe5_4096 libratatosk_hip_wpe5.so 4096
run default2 "" ""
for v in wpe5 wpe6; do RTK_LIB_OVERRIDE=$PWD/ratatosk_amd/variants/libratatosk_hip_$v.so RTK_REGION_WAVES=$([ $v = wpe5 ] && echo 5120 || echo 6144) timeout 600 python -m pytest tests/test_gpu_correct.py -x -q -k "volume or branching" 2>&1 | tail -2 | sed "s/^/$v parity: /" >> $O; done
cat $O

#!/bin/bash
# k_regions by region size class (RTK_TRACE_CLASS: only the regions of one class are run; output not valid, times are) under the default build (128 VGPRs, 4 096 waves) and the
# whole-TU builds at 80 / 72 VGPRs (6 144 / 7 168 waves): do the LIGHT classes gain from more waves per SIMD even though the whole kernel loses?
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/rtk_wd; O=gpurun_out/r06_class_probe.txt; : > $O
B="python bench.py --workdir /tmp/rtk_wd --steps 4 --warmup 1 --no-cpu-baseline --no-host-legs --no-config1-leg --serial"
for cls in 0 1 2 3 4 5 7; do
  for cfg in "default  4096" "wpe6 libratatosk_hip_wpe6.so 6144" "wpe7 libratatosk_hip_wpe7.so 7168"; do set -- $cfg
    ( [ -n "$2" ] && [ "$1" != default ] && export RTK_LIB_OVERRIDE=$PWD/ratatosk_amd/variants/$2 RTK_REGION_WAVES=$3
      RTK_TRACE_CLASS=$cls timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('class $cls $1: k_regions %.2f ms' % d['roofline']['kernel_ms_per_step']['k_regions'])" ) >> $O 2>&1
  done
done
cat $O

#!/bin/bash
# lap profile of k_regions (PROF variant, configs[1]): all searched regions, then only the regions of one gap class (RTK_TRACE_CLASS 0: gap < 40, 3: gap < 256)
# -- where the ~1.1 M cycles go that even the smallest gaps cost
mkdir -p gpurun_out/r04_prof2; W=/tmp/rtk_prof_wd; mkdir -p $W
B="python bench.py --config1-only --steps 2 --warmup 1 --no-cpu-baseline --no-host-legs --serial --workdir $W"
for c in all 0 3; do
  if [ $c = all ]; then unset RTK_TRACE_CLASS; else export RTK_TRACE_CLASS=$c; fi
  RTK_LIB_OVERRIDE=$PWD/ratatosk_amd/variants/libratatosk_hip_prof.so RTK_TRACE=1 timeout 200 $B 2> gpurun_out/r04_prof2/trace_$c.txt | tail -c 300
  echo; echo "== class $c"; grep "lap profile\|k_regions attempt\|fine shares\|by size class" gpurun_out/r04_prof2/trace_$c.txt | tail -4
done

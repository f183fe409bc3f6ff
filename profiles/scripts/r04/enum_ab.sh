#!/bin/bash
# what k_enum's time is made of: kernel stats of the default build and of builds that do the reverse complement / the anchor count loop four times
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_enum_ab; rm -rf $OUT; mkdir -p $OUT; W=/tmp/rtk_enum_wd; mkdir -p $W
B="python bench.py --workdir $W --steps 6 --warmup 2 --no-cpu-baseline --no-host-legs --no-config1-leg --serial"
timeout 300 $B > /dev/null 2> $OUT/warm.err
for v in "" enum_rc4 enum_gap4; do
  lib=ratatosk_amd/libratatosk_hip.so; [ -n "$v" ] && lib=ratatosk_amd/variants/libratatosk_hip_$v.so
  RTK_LIB_OVERRIDE=$PWD/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s_$v -o stats -- $B > /dev/null 2> $OUT/s_$v.err
  f=$(find $OUT/s_$v -name "*kernel_stats.csv" | head -1)
  python - "$f" "[$v]" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0]
    if n in ('k_enum', 'k_stitch', 'k_stitch_copy', 'k_mask', 'k_finalize'): print(sys.argv[2], '%-14s avg %8.1f us min %8.1f max %8.1f' % (n, float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done

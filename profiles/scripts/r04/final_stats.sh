#!/bin/bash
# End of round 4: rocprofv3 kernel trace + stats of the default workload (serial steps) with the final build, and k_mask with segments of 4096 windows
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_final; rm -rf $OUT; mkdir -p $OUT; W=/tmp/rtk_final_wd; mkdir -p $W
B="python bench.py --workdir $W --steps 6 --warmup 2 --no-cpu-baseline --no-host-legs --no-config1-leg --serial"
timeout 300 $B > $OUT/bench_serial.json 2> $OUT/warm.err
python -c "import json; d=json.load(open('$OUT/bench_serial.json')); print('60Mb serial', round(d['value']/1e9,4), round(d['ms_per_step'],2), d['roofline']['kernel_ms_per_step'])"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $B > /dev/null 2> $OUT/stats.err
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_serial.csv; head -14 $OUT/kernel_stats_serial.csv | cut -c1-150
RTK_MASK_SEG=4096 timeout 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('60Mb seg4096', round(d['value']/1e9,4), round(d['ms_per_step'],2), d['roofline']['kernel_ms_per_step'])"
RTK_MASK_SEG=16384 timeout 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('60Mb seg16384', round(d['value']/1e9,4), round(d['ms_per_step'],2), d['roofline']['kernel_ms_per_step'])"

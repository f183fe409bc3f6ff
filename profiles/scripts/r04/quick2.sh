#!/bin/bash
# parity of the stages that changed + kernel stats of the 60 Mb set and the serial line of configs[1]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_quick2; rm -rf $OUT; mkdir -p $OUT; W=/tmp/rtk_q2_wd; mkdir -p $W
timeout 600 python -m pytest tests/test_gpu_seeds.py tests/test_gpu_correct.py tests/test_configs.py -m gpu -x -q 2>&1 | tail -3
B="python bench.py --workdir $W --steps 6 --warmup 2 --no-cpu-baseline --no-host-legs --no-config1-leg --serial"
timeout 300 $B > $OUT/bench_serial.json 2> $OUT/warm.err
python -c "import json; d=json.load(open('$OUT/bench_serial.json')); print('60Mb serial', round(d['value']/1e9,4), round(d['ms_per_step'],2), d['roofline']['kernel_ms_per_step'])"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $B > /dev/null 2> $OUT/stats.err
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_serial.csv
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0]
    if n.startswith('k_') and n != 'k_set_ctx': print('%-16s calls %3s avg %9.1f us min %9.1f max %9.1f' % (n, r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
timeout 300 python bench.py --workdir $W --config1-only --steps 6 --warmup 2 --no-cpu-baseline --no-host-legs --serial 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config1 serial', round(d['value']/1e9,4), round(d['ms_per_step'],2), d['roofline']['kernel_ms_per_step'])"
RTK_TRACE=1 timeout 300 python bench.py --workdir $W --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs --no-config1-leg --serial 2>&1 >/dev/null | grep "finalize, slowest" | tail -1

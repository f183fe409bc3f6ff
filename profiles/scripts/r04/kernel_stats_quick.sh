export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_final; rm -rf $OUT; mkdir -p $OUT; W=/tmp/rtk_final_wd; mkdir -p $W
B="python bench.py --workdir $W --steps 6 --warmup 2 --no-cpu-baseline --no-host-legs --no-config1-leg --serial"
timeout 300 $B > $OUT/bench_serial.json 2> $OUT/warm.err
python -c "import json; d=json.load(open('$OUT/bench_serial.json')); print('60Mb serial', round(d['value']/1e9,4), round(d['ms_per_step'],2), d['roofline']['kernel_ms_per_step'])"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $B > /dev/null 2> $OUT/stats.err
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_serial.csv; python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r04_final/kernel_stats_serial.csv')):
    n=r['Name'].split('(')[0]
    if n.startswith('k_'): print('%-18s calls %3s avg %9.1f us min %9.1f max %9.1f' % (n, r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
grep -c . $W/*.lr.fq 2>/dev/null | head -2; awk 'NR%4==2{ if (length($0)>m) m=length($0) } END{print "longest read", m}' $W/*lr*.fq 2>/dev/null | head

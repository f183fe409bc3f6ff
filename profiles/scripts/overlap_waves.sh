for w in 4096 3072; do
RTK_REGION_WAVES=$w timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('waves $w', round(d['value']/1e6,1), round(d['ms_per_step'],2), d['roofline']['kernel_ms_per_step'])"
done

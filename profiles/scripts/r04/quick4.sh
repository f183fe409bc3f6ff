export TMPDIR=/tmp
W=/tmp/rtk_q4_wd; mkdir -p $W
timeout 300 python -m pytest tests/test_gpu_lookup.py tests/test_gpu_seeds.py tests/test_gpu_correct.py tests/test_toy_golden.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --workdir $W --steps 6 --warmup 2 --no-cpu-baseline --no-host-legs --no-config1-leg --serial 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('60Mb serial', round(d['value']/1e9,4), round(d['ms_per_step'],2), d['roofline']['kernel_ms_per_step'])"
timeout 300 python bench.py --workdir $W --config1-only --steps 6 --warmup 2 --no-cpu-baseline --no-host-legs --serial 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config1 serial', round(d['value']/1e9,4), round(d['ms_per_step'],2), d['roofline']['kernel_ms_per_step'])"

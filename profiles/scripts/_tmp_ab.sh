cd /root/repo
run() { timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs --serial 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"; }
run
run
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('overlapped', d['value'], d['ms_per_step'])"
timeout 600 python -m pytest tests/test_gpu_correct.py tests/test_cli.py -m gpu -x -q 2>&1 | tail -2

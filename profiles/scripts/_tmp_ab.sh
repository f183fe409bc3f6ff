cd /root/repo
timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['host_inclusive']['value'], d['cli_file_to_file']['value'], d['second_pass']['value'])"
timeout 600 python -m pytest tests/test_gpu_seeds.py tests/test_gpu_correct.py tests/test_cli.py tests/test_configs.py -m gpu -x -q 2>&1 | tail -2

#!/bin/bash
# seed stage after the k_mask segments and the k_finalize gather rewrite: parity tests of the stage, then serial kernel times of both sets + the phases of the slowest read
mkdir -p gpurun_out/r04_seeds; W=/tmp/rtk_seeds_wd; mkdir -p $W
timeout 600 python -m pytest tests/test_gpu_seeds.py tests/test_gpu_correct.py tests/test_toy_golden.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r04_seeds/tests.log
RTK_TRACE=1 timeout 200 python bench.py --config1-only --steps 4 --warmup 1 --no-cpu-baseline --no-host-legs --serial --workdir $W 2> gpurun_out/r04_seeds/trace_config1.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config1', round(d['value']/1e9,4), d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
grep "finalize" gpurun_out/r04_seeds/trace_config1.txt | tail -2
RTK_TRACE=1 timeout 400 python bench.py --no-config1-leg --steps 4 --warmup 1 --no-cpu-baseline --no-host-legs --serial --workdir $W 2> gpurun_out/r04_seeds/trace_60mb.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('60Mb', round(d['value']/1e9,4), d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
grep "finalize" gpurun_out/r04_seeds/trace_60mb.txt | tail -2

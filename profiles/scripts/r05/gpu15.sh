cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_graph_load.py tests/test_gpu_seeds.py tests/test_gpu_correct.py tests/test_lanes_regions.py tests/test_myers_lanes.py -x -q -m gpu > gpurun_out/r05_gpu_tests_a.log 2>&1; echo "rc $?" >> gpurun_out/r05_gpu_tests_a.log
tail -4 gpurun_out/r05_gpu_tests_a.log
python bench.py --no-cpu-baseline --no-host-legs --steps 8 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 value %.4g ms_per_step %.2f' % (d['value'], d['ms_per_step'])); print(d['roofline']['kernel_ms_per_step']); print(d['roofline']['k_inexact']); print('c1', d['config1']['value'], d['config1']['ms_per_step'], d['config1']['kernel_ms_per_step'])" > gpurun_out/r05_bench_after_hx.txt 2>&1
cat gpurun_out/r05_bench_after_hx.txt

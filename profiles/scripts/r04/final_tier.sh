#!/bin/bash
# end of round 4: the gpu tests that the last quick runs did not cover (those ran test_gpu_seeds / test_gpu_correct / test_configs at this commit), then the default bench line as the driver runs it
mkdir -p gpurun_out/r04_final_tier
( time timeout 700 python -m pytest tests/test_a2_switch.py tests/test_a3_switch.py tests/test_d1_switch.py tests/test_cli.py tests/test_fixsnps.py tests/test_gpu_lookup.py tests/test_gpu_myers.py tests/test_myers_band.py tests/test_graph_load.py tests/test_index_build.py tests/test_toy_golden.py tests/test_pass2.py -m gpu -x -q 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r04_final_tier/gputier_rest.log
( time timeout 600 python bench.py > gpurun_out/r04_final_tier/bench.json 2> gpurun_out/r04_final_tier/bench.err ) 2>&1 | tail -3 | tee gpurun_out/r04_final_tier/bench_time.txt
tail -c 700 gpurun_out/r04_final_tier/bench.json; tail -3 gpurun_out/r04_final_tier/bench.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2

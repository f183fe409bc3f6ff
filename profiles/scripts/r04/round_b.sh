#!/bin/bash
# device-side table build + device unitigs / colours: the new GPU tests first, then the whole tier, the smoke run and the default bench line
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_round_b; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_graph_load.py tests/test_index_build.py -m gpu -x -q > $OUT/new_tests.log 2>&1; echo "new tests rc=$?"; tail -15 $OUT/new_tests.log
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_tier.log 2>&1; echo "gpu tier rc=$?"; tail -5 $OUT/gpu_tier.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( time timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench_time.txt; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench_time.txt

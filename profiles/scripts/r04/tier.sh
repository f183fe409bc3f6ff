#!/bin/bash
# the whole gpu tier, then the default bench line (as the driver runs it)
mkdir -p gpurun_out/r04_tier
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r04_tier/gputier.log
( time timeout 1500 python bench.py > gpurun_out/r04_tier/bench.json 2> gpurun_out/r04_tier/bench.err ) 2>&1 | tail -3
tail -c 1500 gpurun_out/r04_tier/bench.json; tail -5 gpurun_out/r04_tier/bench.err

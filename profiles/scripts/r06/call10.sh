#!/bin/bash
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_lanes_regions.py -q -m gpu 2>&1 | tail -3 > gpurun_out/r06_late_gpu_tests2.txt; cat gpurun_out/r06_late_gpu_tests2.txt
bash profiles/scripts/r06/class_probe.sh
bash profiles/scripts/r06/pass2_ab.sh
bash profiles/scripts/r06/config4_pmc.sh

#!/bin/bash
# Round 6, final evidence with the build the round ends on: the gpu tests added since the tier run, volume parity beside the tiers (3 x 100 Mb, device against oracle),
# then the profile set (rocprofv3 kernel stats, PMC passes, the bench line itself).
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_lanes_regions.py tests/test_coalesce.py tests/test_capi.py -q -m gpu 2>&1 | tail -3 > gpurun_out/r06_late_gpu_tests.txt
O=gpurun_out/r06_volume_parity.txt; : > $O
timeout 900 python profiles/scripts/parity_volume.py 100 >> $O 2>&1
RTK_LANE_MAX_GAP=128 timeout 900 python profiles/scripts/parity_volume.py 100 2>&1 | sed 's/^volume parity/volume parity (lane kernel on, gap < 128)/' >> $O
timeout 1500 python profiles/scripts/parity_volume.py 100 60000000 0.001 2>&1 | sed 's/^volume parity/volume parity (60 Mb diploid set)/' >> $O
cat gpurun_out/r06_late_gpu_tests.txt; grep "volume parity" $O
bash profiles/scripts/profile_set.sh r06 > gpurun_out/r06_profile_round.log 2>&1
tail -12 gpurun_out/r06_profile_round.log

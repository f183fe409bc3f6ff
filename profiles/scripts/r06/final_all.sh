#!/bin/bash
# Round 6, the evidence of the build the round ends on, in one call: gpu tier + smoke, volume parity beside the tiers (first pass 3 x 100 Mb, second pass 32 Mb), profile set (kernel stats, PMC, the bench line).
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash profiles/scripts/r06/final.sh tier
O=gpurun_out/r06_volume_parity.txt; : > $O
timeout 900 python profiles/scripts/parity_volume.py 100 >> $O 2>&1
RTK_LANE_MAX_GAP=128 timeout 900 python profiles/scripts/parity_volume.py 100 2>&1 | sed 's/^volume parity/volume parity (lane kernel on, gap < 128)/' >> $O
timeout 1500 python profiles/scripts/parity_volume.py 100 60000000 0.001 2>&1 | sed 's/^volume parity/volume parity (60 Mb diploid set)/' >> $O
timeout 900 python profiles/scripts/r05_pass2_volume_parity.py 32 >> $O 2>&1
grep "volume parity" $O
bash profiles/scripts/r06/final.sh prof

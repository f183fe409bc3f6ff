#!/bin/bash
# (1) volume parity beside the test tiers: 100 Mb of configs[1] reads and 100 Mb of reads of the 60 Mb diploid set, device against oracle, the wave kernel alone and with the
# lane-per-region kernel taking its class (RTK_LANE_MAX_GAP=128); (2) k_inexact by its occupancy attribute (variants built by build_variant.sh), serial steps of bench.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_volume_parity.txt; : > $O
timeout 900 python profiles/scripts/parity_volume.py 100 >> $O 2>&1
RTK_LANE_MAX_GAP=128 timeout 900 python profiles/scripts/parity_volume.py 100 2>&1 | sed 's/^volume parity/volume parity (lane kernel on, gap < 128)/' >> $O
timeout 1500 python profiles/scripts/parity_volume.py 100 60000000 0.001 2>&1 | sed 's/^volume parity/volume parity (60 Mb diploid set)/' >> $O
grep "volume parity" $O
W=/tmp/rtk_wpe_wd; mkdir -p $W
for v in "" wpe3 wpe4 wpe8; do
  [ -n "$v" ] && export RTK_LIB_OVERRIDE=$PWD/ratatosk_amd/variants/libratatosk_hip_$v.so || unset RTK_LIB_OVERRIDE
  for rep in 1 2; do timeout 600 python bench.py --workdir $W --steps 8 --warmup 2 --no-cpu-baseline --no-host-legs --serial 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant [$v] c2 ms_per_step %.2f k_inexact %.3f | c1 k_inexact %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step']['k_inexact'], d.get('config1',{}).get('kernel_ms_per_step',{}).get('k_inexact',-1)))"; done
done 2>&1 | tee gpurun_out/r05_inexact_wpe2.txt

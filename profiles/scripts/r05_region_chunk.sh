#!/bin/bash
# k_regions: queue places taken 1 / 2 / 4 / 8 / 16 at a time (RTK_REGION_CHUNK), serial steps of bench.py on the 60 Mb set and configs[1]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=/tmp/rtk_chunk_wd; mkdir -p $W
for c in 1 2 4 8 16 1 4; do
  RTK_REGION_CHUNK=$c timeout 600 python bench.py --workdir $W --steps 8 --warmup 2 --no-cpu-baseline --no-host-legs --serial 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $c: c2 ms_per_step %.2f k_regions %.3f | c1 ms_per_step %.2f k_regions %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step']['k_regions'], d['config1']['ms_per_step'], d['config1']['kernel_ms_per_step']['k_regions']))"
done

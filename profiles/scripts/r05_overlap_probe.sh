#!/bin/bash
# Developer run: does running consecutive steps on two streams pay once the persistent region kernel leaves wave slots free? (configs[1], 64 Mb steps)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --config1-only --no-cpu-baseline --no-host-legs --steps 10 --warmup 2"
( for RW in 4096 3584 3072 2560; do
  for M in "" "--overlap"; do
    echo "== RTK_REGION_WAVES=$RW $M"; RTK_REGION_WAVES=$RW timeout 900 $B $M 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.4g ms_per_step %.2f' % (d['value'], d['ms_per_step']))"
  done; done ) > gpurun_out/r05_overlap_probe.txt 2>&1
cat gpurun_out/r05_overlap_probe.txt

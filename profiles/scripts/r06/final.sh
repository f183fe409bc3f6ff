#!/bin/bash
# the driver's GPU tier (tests + smoke), then the counter / kernel-stat passes and the bench line of the round with the same build.   usage: r06_final.sh [tier|prof|all]
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
W=${1:-all}
if [ "$W" = tier ] || [ "$W" = all ]; then
  ( time timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -40 ) > gpurun_out/r06_gpu_tier.log 2>&1
  ( time timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5 ) >> gpurun_out/r06_gpu_tier.log 2>&1
  tail -14 gpurun_out/r06_gpu_tier.log
fi
if [ "$W" = prof ] || [ "$W" = all ]; then
  bash profiles/scripts/profile_set.sh r06 > gpurun_out/r06_profile_round.log 2>&1
  tail -12 gpurun_out/r06_profile_round.log
fi

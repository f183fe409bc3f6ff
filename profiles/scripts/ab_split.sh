#!/bin/bash
# A/B on one box: the library as built against a variant .so (RTK_LIB_OVERRIDE), per-kernel times with one step at a time; parity of the variant (developer)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ALT=${1:-ratatosk_amd/libratatosk_hip_split.so}
for rep in 1 2; do
  echo "base:"; bash profiles/scripts/quick_serial.sh
  echo "variant $ALT:"; RTK_LIB_OVERRIDE=$PWD/$ALT bash profiles/scripts/quick_serial.sh
done
echo "config2 graph, base / variant:"; bash profiles/scripts/quick_serial.sh --config2; RTK_LIB_OVERRIDE=$PWD/$ALT bash profiles/scripts/quick_serial.sh --config2
RTK_LIB_OVERRIDE=$PWD/$ALT timeout 900 python -m pytest tests/test_gpu_correct.py tests/test_gpu_seeds.py tests/test_toy_golden.py -x -q -m gpu 2>&1 | tail -2

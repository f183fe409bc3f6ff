#!/bin/bash
# per-kernel times of the default workload, one step at a time (developer A/B): python bench.py --serial, 4 steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-legs --serial $@ 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), d['roofline']['kernel_ms_per_step'])"

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RTK_MYERS_PROF=1
timeout 300 python profiles/scripts/band_timing.py 10000 2>&1 | grep -v "driver 0 cycles" | tail -8

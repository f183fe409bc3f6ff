cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(time timeout 2400 python -m pytest tests -x -q -m gpu --durations=12) > gpurun_out/gputier.log 2>&1; echo "rc=$?" >> gpurun_out/gputier.log
tail -25 gpurun_out/gputier.log

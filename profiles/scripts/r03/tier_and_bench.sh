#!/bin/bash
# the gpu test tier, then the default bench line (developer; every step under its own timeout)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(time timeout 2400 python -m pytest tests -x -q -m gpu --durations=8) > gpurun_out/gputier.log 2>&1; echo "rc=$?" >> gpurun_out/gputier.log
tail -16 gpurun_out/gputier.log
(time timeout 900 python bench.py) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["host_inclusive"]["value"], d["cli_file_to_file"], d["second_pass"]["value"], d["config"]["setup_s"])
PY

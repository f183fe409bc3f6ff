#!/bin/bash
# end of round 3: rocprofv3 kernel stats + PMC passes + bench line (profile_round.sh), then the gpu test tier
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2700 bash profiles/scripts/profile_round.sh r03 > gpurun_out/profile_round.log 2>&1; echo "profile rc=$?"
tail -4 gpurun_out/profile_round.log
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/profiles_r03/r03_bench.json").read().strip().splitlines()[-1])
    print("value %.3e ms/step %.2f frac %s traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"]))
    print("serial", d["roofline"].get("kernel_ms_per_step_serial"))
    print("host_inclusive", d.get("host_inclusive", {}).get("value")); print("cli", d.get("cli_file_to_file", {}).get("value")); print("second_pass", d.get("second_pass")); print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("parity_on_sample"))
except Exception as e: print("parse failed", e)
PY
(time timeout 2400 python -m pytest tests -x -q -m gpu --durations=5) > gpurun_out/gputier.log 2>&1; echo "tier rc=$?"
tail -12 gpurun_out/gputier.log

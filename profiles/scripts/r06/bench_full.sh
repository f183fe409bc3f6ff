#!/bin/bash
# the driver's command (python bench.py, no flags) on one MI355X, timed; the whole JSON line is kept
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err ) 2> gpurun_out/r06_bench_time.txt
tail -3 gpurun_out/r06_bench.err; cat gpurun_out/r06_bench_time.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench.json").read().strip().splitlines()[-1])
print("value %.4g ms_per_step %.2f value_correction_phase %s" % (d["value"], d["ms_per_step"], d.get("value_correction_phase")))
print("kernels", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"], "k_inexact frac", d["roofline"]["k_inexact"]["frac"])
print("host_inclusive", d.get("host_inclusive", {}).get("value"), "cli", d.get("cli_file_to_file", {}).get("value"), "second_pass", d.get("second_pass", {}).get("value"))
for k, v in d.get("by_ticket_size", {}).items():
    if isinstance(v, dict): print(k, {c: ("%.3g" % v[c] if isinstance(v[c], float) else v[c]) for c in v if "callers" in c})
print("config1", d.get("config1", {}).get("value"), d.get("config1", {}).get("ms_per_step"))
c4 = d.get("config4", {}); print("config4", {k: c4.get(k) for k in ("skipped", "error", "tickets", "build_index_s")}, c4.get("graph", {}).get("hbm_gb"))
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("parity_on_sample"))
PY

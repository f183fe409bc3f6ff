#!/bin/bash
# end of round 3: volume parity outside the test tiers (HIP path vs oracle on all host threads), latency of one whole-read alignment by workgroup size
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python profiles/scripts/parity_volume.py 96 2>&1 | tail -3
timeout 1200 python profiles/scripts/parity_volume.py 32 60000000 0.001 2>&1 | tail -3
rm -rf gpurun_out/bt; timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bt -o bt -- python profiles/scripts/band_timing.py 10000 24000 40000 98000 2>&1 | grep "^len"
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/bt/**/bt_kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0]
    if n.startswith("k_myers"): print(n, r.get("Workgroup_Size_X", r.get("Workgroup_Size")), "%.2f ms" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY

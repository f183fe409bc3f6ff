cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_myers_band.py tests/test_gpu_myers.py -x -q -m gpu > gpurun_out/band_tests.log 2>&1; echo "rc=$?" >> gpurun_out/band_tests.log
tail -5 gpurun_out/band_tests.log
export RTK_MYERS_PROF=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/band_prof -o bt -- python profiles/scripts/band_timing.py > gpurun_out/band_timing.log 2>&1
grep -v "^[WE]2026" gpurun_out/band_timing.log | tail -24
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/band_prof/**/bt_kernel_trace.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if d > 0.05: print("%-28s %9.2f ms  wg %s" % (r["Kernel_Name"].split("(")[0], d, r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
PY

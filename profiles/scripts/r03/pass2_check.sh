cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_pass2.py tests/test_fixsnps.py -x -q -m gpu > gpurun_out/pass2_tests.log 2>&1; echo "rc=$?" >> gpurun_out/pass2_tests.log
tail -5 gpurun_out/pass2_tests.log
timeout 1500 python profiles/scripts/pass2_rate.py 5e6 ${1:-128e6} 63 > gpurun_out/pass2_rate.json 2> gpurun_out/pass2_rate.err; echo "rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/pass2_rate.json"))
print(d["pass1"]); print(d["pass2"])
for l in d["pass2_trace_head"]:
    if "phase attempt" in l or "wave-0" in l or "k_regions attempt" in l or "seeds attempt" in l: print(l)
PY

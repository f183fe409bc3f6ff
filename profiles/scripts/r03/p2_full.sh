cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_pass2.py tests/test_fixsnps.py tests/test_myers_band.py tests/test_gpu_myers.py tests/test_gpu_correct.py -x -q -m gpu > gpurun_out/p2_tests.log 2>&1; echo "rc=$?" >> gpurun_out/p2_tests.log
tail -4 gpurun_out/p2_tests.log
timeout 1500 python bench.py --steps 6 --warmup 2 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_quick.json").read().strip().splitlines()[-1])
    print("bench value %.3e ms/step %.2f" % (d["value"], d["ms_per_step"])); print(d["roofline"].get("kernel_ms_per_step")); print("second_pass", d.get("second_pass")); print("cli", d.get("cli_file_to_file"))
except Exception as e: print("bench parse failed", e)
PY

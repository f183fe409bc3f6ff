cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RTK_PHASE_LONG=24576 RTK_PHASE_LGRID=64 RTK_PHASE_MGRID=128
for w in 3 5 7; do
  timeout 1200 python profiles/scripts/pass2_rate.py 5e6 256e6 63 --workers-per-gpu $w > gpurun_out/pass2_w$w.json 2> gpurun_out/pass2_w$w.err; echo "rc=$?"
  python - <<PY
import json
d = json.load(open("gpurun_out/pass2_w$w.json"))
print("workers $w", d["pass2"])
ph = [l for l in d["pass2_trace_head"] if "phase attempt" in l]
for l in ph[:4]: print("   ", l[:200])
PY
done

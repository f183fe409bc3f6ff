cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RTK_P2_AB="${RTK_P2_AB:-RTK_PHASE_LONG=16384;RTK_PHASE_LONG=32768;RTK_PHASE_LONG=0}"
timeout 2500 python profiles/scripts/pass2_rate.py 5e6 ${1:-128e6} 63 > gpurun_out/pass2_ab.json 2> gpurun_out/pass2_ab.err; echo "rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/pass2_ab.json"))
print("pass1", d["pass1"])
print("base", d["pass2"])
for l in d["pass2_trace_head"]:
    if "phase attempt" in l or "wave-0" in l or "k_regions attempt" in l or "seeds attempt" in l: print("   ", l[:230])
for a in d["ab"]:
    print(a["env"], a["pass2"])
    for l in a["phase"][:6]: print("   ", l[:230])
PY
timeout 900 python bench.py --steps 6 --warmup 2 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_quick.json").read().strip().splitlines()[-1])
    print("bench value %.3e ms/step %.2f" % (d["value"], d["ms_per_step"])); print(d["roofline"].get("kernel_ms_per_step")); print("second_pass", d.get("second_pass"))
except Exception as e: print("bench parse failed", e)
PY

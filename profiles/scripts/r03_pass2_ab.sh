cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RTK_P2_AB="${RTK_P2_AB:-RTK_PHASE_LONG=16384;RTK_PHASE_LONG=24000;RTK_PHASE_LONG=16384,RTK_PHASE_LGRID=128,RTK_PHASE_MGRID=256;RTK_PHASE_LONG=32768}"
timeout 2500 python profiles/scripts/pass2_rate.py 5e6 ${1:-128e6} 63 > gpurun_out/pass2_ab.json 2> gpurun_out/pass2_ab.err; echo "rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/pass2_ab.json"))
print("base", d["pass2"])
for a in d["ab"]:
    print(a["env"], a["pass2"])
    for l in a["phase"][:6]: print("   ", l[:200])
PY

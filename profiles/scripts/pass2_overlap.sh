#!/bin/bash
# steady-state second pass under rocprofv3 --kernel-trace: per-kernel durations and how many kernels run side by side (developer)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python profiles/scripts/pass2_rate.py 5e6 128e6 63 > /tmp/p2rate.json 2>/tmp/p2rate.err
WD=$(ls -d /tmp/rtk_p2_* | tail -1)
for i in 1 2 3 4 5 6; do echo $WD/c2.2.fastq >> $WD/in.txt; echo $WD/c2.lr.fq >> $WD/raw.txt; done
rm -rf gpurun_out/p2o
RTK_CLI_STATS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p2o -o p2o -- ratatosk_amd/bin/Ratatosk correct -2 -K 63 -c 16 --workers-per-gpu ${1:-6} -g $WD/c2.p2.index.k63.fasta.gz -d $WD/c2.p2.index.k63.rtsk -l $WD/in.txt -L $WD/raw.txt -o $WD/again 2>&1 | grep "correction phase"
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/p2o/**/p2o_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = []; by = collections.defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); n = r["Kernel_Name"].split("(")[0]
    wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "64")); g = r.get("Grid_Size_X", r.get("Grid_Size", "0"))
    key = n + ("/" + wg if n.startswith("k_phase") else "") + ("/g" + g if n == "k_phase" else "")
    by[key].append((e - s) / 1e6); ev.append((s, 1)); ev.append((e, -1))
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:12]: print("%-28s n=%4d avg %8.2f ms  max %8.2f  total %9.1f ms" % (k, len(v), sum(v) / len(v), max(v), sum(v)))
ev.sort(); cur = 0; last = ev[0][0]; hist = collections.defaultdict(float)
for t, d in ev:
    hist[cur] += (t - last) / 1e6; last = t; cur += d
tot = sum(hist.values()); print("span %.1f ms; time with N kernels running:" % tot, {k: round(v, 1) for k, v in sorted(hist.items())})
PY

#!/bin/bash
# kernel durations of one second-pass run (developer): data + indexes like pass2_rate.py, then `correct -2` under rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python profiles/scripts/pass2_rate.py 5e6 ${2:-64e6} 63 > /tmp/p2rate.json 2>/tmp/p2rate.err
WD=$(ls -d /tmp/rtk_p2_* | tail -1)
rm -rf gpurun_out/p2k
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p2k -o p2k -- ratatosk_amd/bin/Ratatosk correct -2 -K 63 -c 16 --workers-per-gpu ${1:-1} -g $WD/c2.p2.index.k63.fasta.gz -d $WD/c2.p2.index.k63.rtsk -l $WD/c2.2.fastq -L $WD/c2.lr.fq -o $WD/again > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/p2k/**/p2k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > 1.0: print("%-14s start %8.1f ms  dur %8.1f ms  grid %s wg %s" % (r["Kernel_Name"].split("(")[0], (int(r["Start_Timestamp"]) - t0) / 1e6, d, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
PY

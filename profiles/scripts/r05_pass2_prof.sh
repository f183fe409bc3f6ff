#!/bin/bash
# second pass, where the time of a ticket goes (round 5): data + indexes by pass2_rate.py (which also prints the file-to-file rate with the default tickets in flight and the
# trace of the first tickets), then ONE ticket at a time under rocprofv3 --kernel-trace: every kernel above 0.5 ms with its start, duration and grid.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_pass2; mkdir -p $OUT
timeout 900 python profiles/scripts/pass2_rate.py 5e6 ${1:-128e6} 63 > $OUT/p2rate.json 2> $OUT/p2rate.err
WD=$(ls -d /tmp/rtk_p2_* | tail -1)
python - <<PY
import json; d = json.load(open("$OUT/p2rate.json")); print(d["pass2"]); print("\n".join(d["pass2_trace_head"][:14]))
PY
rm -rf $OUT/p2k
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/p2k -o p2k -- ratatosk_amd/bin/Ratatosk correct -2 -K 63 -c 16 --workers-per-gpu 1 -g $WD/c2.p2.index.k63.fasta.gz -d $WD/c2.p2.index.k63.rtsk -l $WD/c2.2.fastq -L $WD/c2.lr.fq -o $WD/again > /dev/null 2>&1
python - <<PY > $OUT/kernels_one_ticket_at_a_time.txt
import csv, glob
f = glob.glob("$OUT/p2k/**/p2k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
tot = {}
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; nm = r["Kernel_Name"].split("(")[0]
    tot[nm] = tot.get(nm, 0) + d
    if d > 0.5: print("%-22s start %9.1f ms  dur %8.2f ms  grid %s wg %s" % (nm[:22], (int(r["Start_Timestamp"]) - t0) / 1e6, d, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
print("totals (ms):", sorted(((round(v, 1), k[:30]) for k, v in tot.items()), reverse=True)[:14])
PY
head -70 $OUT/kernels_one_ticket_at_a_time.txt; tail -1 $OUT/kernels_one_ticket_at_a_time.txt
find $OUT/p2k -name "*.csv" -size +4M -delete

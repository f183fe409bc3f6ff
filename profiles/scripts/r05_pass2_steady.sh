#!/bin/bash
# second pass in steady state under a kernel trace (round 5): how full is the machine? data + indexes by pass2_rate.py, then the files COPIES times through `correct -2`
# with the default tickets in flight under rocprofv3 --kernel-trace; prints the rate, the share of the wall with no kernel running, wave-slot use by kernel.
#   usage: r05_pass2_steady.sh [REF=5e6] [LR=128e6] [COPIES=10] [extra CLI options...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
REF=${1:-5e6}; LR=${2:-128e6}; COPIES=${3:-10}; shift 3
OUT=gpurun_out/r05_pass2; mkdir -p $OUT
timeout 1500 python profiles/scripts/pass2_rate.py $REF $LR 63 > $OUT/p2rate_steady.json 2> $OUT/p2rate_steady.err
WD=$(ls -d /tmp/rtk_p2_* | tail -1)
rm -f $WD/in.txt $WD/raw.txt; for i in $(seq $COPIES); do echo $WD/c2.2.fastq >> $WD/in.txt; echo $WD/c2.lr.fq >> $WD/raw.txt; done
CMD="ratatosk_amd/bin/Ratatosk correct -2 -K 63 -c 16 -g $WD/c2.p2.index.k63.fasta.gz -d $WD/c2.p2.index.k63.rtsk -l $WD/in.txt -L $WD/raw.txt -o $WD/again $*"
echo "== untraced"; RTK_CLI_STATS=1 timeout 600 $CMD 2>&1 | grep "correction phase"
[ -n "${RTK_P2_AB:-}" ] && { IFS=';' read -ra SPECS <<< "$RTK_P2_AB"; for s in "${SPECS[@]}"; do echo "== $s"; env $(echo $s | tr ',' ' ') RTK_CLI_STATS=1 timeout 600 $CMD 2>&1 | grep "correction phase"; done; }
rm -rf $OUT/p2s
RTK_CLI_STATS=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/p2s -o p2s -- $CMD 2>&1 | grep "correction phase"
python - <<PY | tee $OUT/steady_state_kernels.txt
import csv, glob
f = glob.glob("$OUT/p2s/**/p2s_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)) for r in csv.DictReader(open(f))]
rows = [r for r in rows if r[2] not in ("k_tables_insert", "k_hx_scatter", "k_hx_mark", "k_hx_keys", "k_fill_slots", "k_tables_adjacency") and "rocprim" not in r[2]]
big = [r for r in rows if r[2] in ("k_phase", "k_phase_long", "k_regions")]
t0, t1 = min(r[0] for r in big), max(r[1] for r in big)
ev = sorted([(r[0], 1) for r in rows] + [(r[1], -1) for r in rows])
idle, cur, last, conc = 0, 0, t0, 0.0
for t, d in ev:
    if t > t0 and t <= t1:
        if cur == 0: idle += t - max(last, t0)
        conc += cur * (t - max(last, t0))
    cur += d; last = t
wall = (t1 - t0) / 1e6
print("window of the correction kernels %.1f ms; no kernel running %.1f %% of it; mean number of kernels running %.2f" % (wall, 100.0 * idle / 1e6 / wall, conc / 1e6 / wall))
tot = {}
for s, e, nm, g in rows:
    w = min(g // 64, 4096); a = tot.setdefault(nm, [0, 0.0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e6; a[2] += w * (e - s) / 1e6
print("kernel: launches, sum of durations (ms), wave-slot x ms as launched (waves capped at 4096) and as a share of 4096 x window")
for nm, (n, d, ws) in sorted(tot.items(), key=lambda x: -x[1][2])[:10]:
    print("  %-22s %5d %10.1f %12.0f  %5.1f %%" % (nm[:22], n, d, ws, 100.0 * ws / (4096.0 * wall)))
PY
find $OUT/p2s -name "*.csv" -size +4M -delete

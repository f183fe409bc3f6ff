cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python profiles/scripts/pass2_rate.py 60e6 128e6 63 > /tmp/p2.json 2> /tmp/p2.err
WD=$(ls -d /tmp/rtk_p2_* | tail -1)
rm -f $WD/in.txt $WD/raw.txt; for i in $(seq 18); do echo $WD/c2.2.fastq >> $WD/in.txt; echo $WD/c2.lr.fq >> $WD/raw.txt; done
for r in 1 2 3 4 5; do sleep 3; echo "== run $r"; RTK_CLI_STATS=1 RTK_CLI_TRACE=1 RTK_TRACE=1 timeout 300 ratatosk_amd/bin/Ratatosk correct -2 -K 63 -c 16 -g $WD/c2.p2.index.k63.fasta.gz -d $WD/c2.p2.index.k63.rtsk -l $WD/in.txt -L $WD/raw.txt -o $WD/again > /tmp/p2run.log 2>&1
grep "correction phase" /tmp/p2run.log | sed 's/thread-seconds.*//'; grep -c "phase_take" /tmp/p2run.log; grep "phase_take" /tmp/p2run.log | head -12; grep "attempt [12]" /tmp/p2run.log | head -3
python - <<'PY'
import re
tot=[]
for l in open("/tmp/p2run.log"):
    m = re.search(r"start \+([0-9.]+) ms, create ([0-9.]+), run ([0-9.]+), fetch ([0-9.]+)", l)
    if m: tot.append(tuple(float(x) for x in m.groups()))
tot.sort()
print("create/run by ticket order:", " ".join("%d/%d" % (t[1], t[2]) for t in tot[:70:3]))
PY
done

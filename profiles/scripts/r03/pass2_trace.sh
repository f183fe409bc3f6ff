cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python profiles/scripts/pass2_rate.py 5e6 128e6 63 > gpurun_out/pass2_t.json 2> gpurun_out/pass2_t.err; echo "rc=$?"
WD=$(ls -d /tmp/rtk_p2_* | tail -1)
for i in 1 2 3 4 5 6; do echo $WD/c2.2.fastq >> $WD/in.txt; echo $WD/c2.lr.fq >> $WD/raw.txt; done
for W in ${WORKERS:-6}; do
RTK_TRACE=1 RTK_CLI_STATS=1 ratatosk_amd/bin/Ratatosk correct -2 -K 63 -c 16 --workers-per-gpu $W -g $WD/c2.p2.index.k63.fasta.gz -d $WD/c2.p2.index.k63.rtsk -l $WD/in.txt -L $WD/raw.txt -o $WD/again > /dev/null 2> gpurun_out/pass2_trace_w$W.txt
python - <<PY
import re
t = open("gpurun_out/pass2_trace_w$W.txt").read()
ph = [float(x) for x in re.findall(r"phase attempt 0: ([0-9.]+) ms", t)]
rg = [float(x) for x in re.findall(r"k_regions attempt 0: ([0-9.]+) ms", t)]
rr = [float(x) for x in re.findall(r"regions done\s+\+([0-9.]+) ms", t)]
sd = [float(x) for x in re.findall(r"seeds attempt 0: ([0-9.]+) ms", t)]
avg = lambda v: sum(v) / max(1, len(v))
print("workers $W: tickets %d  phase avg %.1f max %.1f | seeds avg %.1f | k_regions avg %.1f max %.1f | regions stage avg %.1f" % (len(ph), avg(ph), max(ph), avg(sd), avg(rg), max(rg), avg(rr)))
print([l for l in t.splitlines() if "correction phase" in l])
PY
done

#!/bin/bash
# second pass: per-ticket times of the workers (create = pack + H2D, run = phasing + seeds + regions, fetch) with the default tickets in flight, and the stages inside run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python profiles/scripts/pass2_rate.py ${1:-60e6} 128e6 63 > /tmp/p2.json 2> /tmp/p2.err
WD=$(ls -d /tmp/rtk_p2_* | tail -1)
rm -f $WD/in.txt $WD/raw.txt; for i in $(seq 10); do echo $WD/c2.2.fastq >> $WD/in.txt; echo $WD/c2.lr.fq >> $WD/raw.txt; done
RTK_CLI_STATS=1 RTK_CLI_TRACE=1 RTK_TRACE=1 timeout 300 ratatosk_amd/bin/Ratatosk correct -2 -K 63 -c 16 -g $WD/c2.p2.index.k63.fasta.gz -d $WD/c2.p2.index.k63.rtsk -l $WD/in.txt -L $WD/raw.txt -o $WD/again > /tmp/p2run.log 2>&1
grep "correction phase" /tmp/p2run.log
grep "cli trace" /tmp/p2run.log | sed -n 12,40p
python - <<'PY'
import re
ph, sd, rg, tot = [], [], [], []
for l in open("/tmp/p2run.log"):
    m = re.search(r"phase attempt 0: ([0-9.]+) ms", l);  ph += [float(m.group(1))] if m else []
    m = re.search(r"seeds attempt 0: ([0-9.]+) ms", l);  sd += [float(m.group(1))] if m else []
    m = re.search(r"k_regions attempt 0: ([0-9.]+) ms", l); rg += [float(m.group(1))] if m else []
    m = re.search(r"create ([0-9.]+), run ([0-9.]+), fetch ([0-9.]+)", l); tot += [tuple(float(x) for x in m.groups())] if m else []
av = lambda v: sum(v) / max(1, len(v))
print("tickets %d: phase %.1f ms, seeds %.1f, k_regions %.1f | worker: create %.1f, run %.1f, fetch %.1f" % (len(tot), av(ph), av(sd), av(rg), av([t[0] for t in tot]), av([t[1] for t in tot]), av([t[2] for t in tot])))
PY
grep "regions alloc\|regions enum\|regions regions\|regions done\|phase pack\|repack" /tmp/p2run.log | head -12

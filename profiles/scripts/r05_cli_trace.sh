cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=/tmp/rtk_cliab; mkdir -p $W
python - <<PY
import sys; sys.path.insert(0, ".")
import bench
pre = bench.make_dataset("$W", 60_000_000, 140_000_000, snps=True, het=0.001)
one = open(pre + ".lr.fq", "rb").read()
with open("$W/in.fq", "wb") as f:
    for _ in range(max(2, min(48, int(7.2e9 // max(1, len(one) // 2))))): f.write(one)
PY
PRE=$W/c2
for r in 1 2; do RTK_TRACE=1 RTK_CLI_TRACE=1 RTK_CLI_STATS=1 timeout 600 ratatosk_amd/bin/Ratatosk correct -1 -c 16 --gpus 1 -g $PRE.index.k31.fasta.gz -d $PRE.index.k31.rtsk -l $W/in.fq -o $W/out > /tmp/clirun.log 2>&1
grep "correction phase" /tmp/clirun.log | sed 's/^.*correction phase/correction phase/; s/thread-seconds.*//'; echo "stage_take misses $(grep -c stage_take /tmp/clirun.log)"; grep "rtk trace\] fetch" /tmp/clirun.log | sed -n "40,52p"; echo "pool_take $(grep -c pool_take /tmp/clirun.log) pool_give-free $(grep -c pool_give /tmp/clirun.log)"; grep "pool_take" /tmp/clirun.log | awk '{s+=$(NF-1); if ($(NF-1)>m) m=$(NF-1)} END {print "ms in pool_take mallocs:", s, "max", m}'
python - <<'PY'
import re
t=[]
for l in open("/tmp/clirun.log"):
    m=re.search(r"start \+([0-9.]+) ms, create ([0-9.]+), run ([0-9.]+), fetch ([0-9.]+)", l)
    if m: t.append(tuple(float(x) for x in m.groups()))
t.sort(); print("create/run/fetch:", " ".join("%d/%d/%d"%(a[1],a[2],a[3]) for a in t[::7]))
PY
done

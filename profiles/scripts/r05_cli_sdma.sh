#!/bin/bash
# the 22 ms D2H of a ticket's records (3 GB/s; the others 1.3 ms = 51 GB/s): copy engine or copy kernel?  CLI file to file on the 60 Mb set under environment settings
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
run() { echo "== $*"; env "$@" RTK_TRACE=1 RTK_CLI_STATS=1 timeout 600 ratatosk_amd/bin/Ratatosk correct -1 -c 16 --gpus 1 -g $PRE.index.k31.fasta.gz -d $PRE.index.k31.rtsk -l $W/in.fq -o $W/out > /tmp/clirun.log 2>&1
grep "correction phase" /tmp/clirun.log | sed 's/^.*correction phase/correction phase/; s/thread-seconds.*//'
grep "rtk trace\] fetch" /tmp/clirun.log | awk '{ms=$(NF-1); if (ms > 8) slow++; else fast++; s+=ms} END {print "fetches: fast", fast, "slow", slow, "mean ms", s/(fast+slow)}'; }
run A=0
for rep in 1 2; do
run A=0
run HSA_ENABLE_SDMA=0
run GPU_MAX_HW_QUEUES=8
run HIP_FORCE_DEV_KERNARG=1
done

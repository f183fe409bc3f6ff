#!/bin/bash
# first pass through the CLI, file to file on the 60 Mb set: bases per ticket (-B) and tickets in flight (--workers-per-gpu); every setting twice
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=/tmp/rtk_cliab; mkdir -p $W
python - <<PY
import sys; sys.path.insert(0, ".")
import bench
pre = bench.make_dataset("$W", 60_000_000, 140_000_000, snps=True, het=0.001)
one = open(pre + ".lr.fq", "rb").read()
with open("$W/in.fq", "wb") as f:
    for _ in range(max(2, min(48, int(7.2e9 // max(1, len(one) // 2))))): f.write(one)
print(pre)
PY
PRE=$W/c2
run() { echo "== $*"; RTK_CLI_STATS=1 timeout 600 ratatosk_amd/bin/Ratatosk correct -1 -c 16 --gpus 1 -g $PRE.index.k31.fasta.gz -d $PRE.index.k31.rtsk -l $W/in.fq -o $W/out "$@" 2>&1 | grep "correction phase" | sed 's/^.*correction phase/correction phase/'; }
run > /dev/null
for rep in 1 2; do
run
run -B 50331648
run -B 67108864
run -B 67108864 --workers-per-gpu 4
run --workers-per-gpu 4
done

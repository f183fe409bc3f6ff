#!/bin/bash
# per-ticket timeline of the first-pass CLI (developer)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WD=$(mktemp -d /tmp/rtk_io_XXXX)
python - <<PY
import sys; sys.path.insert(0, ".")
import bench
bench.make_dataset("$WD", 5000000, 150000000, snps=True)
PY
for i in 1 2 3 4 5 6; do echo $WD/c2.lr.fq >> $WD/plain.txt; done
RTK_CLI_TRACE=1 RTK_TRACE=1 RTK_SERIAL_READER=${SERIAL:-1} RTK_CLI_STATS=1 timeout 300 ratatosk_amd/bin/Ratatosk correct -1 -c 16 --workers-per-gpu ${W:-4} -B ${B:-64000000} -g $WD/c2.index.k31.fasta.gz -d $WD/c2.index.k31.rtsk -l $WD/plain.txt -o $WD/out 2>&1 | grep -v "size class\|fine shares\|shares of\|cycle shares\|chooseColors\|walks \|finalize wave" | head -${N:-70}
rm -rf $WD

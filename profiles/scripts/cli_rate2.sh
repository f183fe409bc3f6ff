#!/bin/bash
# first-pass CLI at steady state: 24 copies of the long-read file (3.5 Gb), ticket sizes / workers (developer)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WD=$(mktemp -d /tmp/rtk_io_XXXX)
python - <<PY
import sys; sys.path.insert(0, ".")
import bench
bench.make_dataset("$WD", 5000000, 150000000, snps=True)
PY
for i in $(seq 24); do echo $WD/c2.lr.fq >> $WD/plain.txt; done
for V in ${VARIANTS:-"3 64000000" "4 64000000" "3 100000000" "4 100000000" "3 48000000"}; do set -- $V
  RTK_CLI_STATS=1 timeout 300 ratatosk_amd/bin/Ratatosk correct -1 -c 16 --workers-per-gpu $1 -B $2 -g $WD/c2.index.k31.fasta.gz -d $WD/c2.index.k31.rtsk -l $WD/plain.txt -o $WD/out 2>&1 | grep "correction phase" | sed "s/^/workers $1 -B $2: /"
done
rm -rf $WD

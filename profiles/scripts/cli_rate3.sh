#!/bin/bash
# first-pass CLI at steady state with formatter threads: ONE 7 GB file (the generated long reads 24 times, 3.5 Gb), workers / -c; reader alone (developer)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WD=$(mktemp -d /tmp/rtk_io_XXXX)
python - <<PY
import sys; sys.path.insert(0, ".")
import bench
bench.make_dataset("$WD", 5000000, 150000000, snps=True)
PY
for i in $(seq 24); do cat $WD/c2.lr.fq >> $WD/big.fq; done
ls -la $WD/big.fq | awk '{print "FASTQ bytes:", $5}'
for C in 1 4 16 32 64; do timeout 120 ratatosk_amd/bin/Ratatosk correct -1 --parse-only -c $C -B 64000000 -l $WD/big.fq 2>&1 | tail -1 | sed "s/^/parse-only -c $C: /"; done
for V in "3 16 64000000" "4 16 64000000" "3 32 64000000" "3 16 96000000" "3 16 64000000"; do set -- $V
  RTK_CLI_STATS=1 timeout 300 ratatosk_amd/bin/Ratatosk correct -1 -c $2 --workers-per-gpu $1 -B $3 -g $WD/c2.index.k31.fasta.gz -d $WD/c2.index.k31.rtsk -l $WD/big.fq -o $WD/out 2>&1 | grep "correction phase" | sed "s/^/workers $1 -c $2 -B $3: /"
done
RTK_CLI_TRACE=1 RTK_CLI_STATS=1 timeout 300 ratatosk_amd/bin/Ratatosk correct -1 -c 16 -g $WD/c2.index.k31.fasta.gz -d $WD/c2.index.k31.rtsk -l $WD/big.fq -o $WD/out 2> gpurun_out/cli_trace3.txt; tail -1 gpurun_out/cli_trace3.txt
rm -rf $WD

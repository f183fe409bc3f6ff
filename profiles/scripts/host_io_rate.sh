#!/bin/bash
# host I/O of the first pass (developer measurement; DESIGN_HISTORY.md section 5): reader alone, CLI file to file on plain and on gzipped input
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WD=$(mktemp -d /tmp/rtk_io_XXXX)
python - <<PY
import sys; sys.path.insert(0, ".")
import bench
bench.make_dataset("$WD", 5000000, 150000000, snps=True)
PY
EXE=ratatosk_amd/bin/Ratatosk
for i in 1 2 3 4 5 6; do echo $WD/c2.lr.fq >> $WD/plain.txt; done
ls -la $WD/c2.lr.fq | awk '{print "long-read FASTQ bytes:", $5}'
for C in 1 4 16 32 64; do timeout 120 $EXE correct -1 --parse-only -c $C -B 64000000 -l $WD/plain.txt; done
timeout 120 env RTK_SERIAL_READER=1 $EXE correct -1 --parse-only -c 1 -l $WD/plain.txt >/dev/null
for W in 3 4 6; do
RTK_CLI_STATS=1 timeout 300 $EXE correct -1 -c 16 --workers-per-gpu $W -B 64000000 -g $WD/c2.index.k31.fasta.gz -d $WD/c2.index.k31.rtsk -l $WD/plain.txt -o $WD/out 2>&1 | grep "correction phase" | sed "s/^/plain, parallel reader, workers $W: /"
done
RTK_SERIAL_READER=1 RTK_CLI_STATS=1 timeout 300 $EXE correct -1 -c 16 -B 64000000 -g $WD/c2.index.k31.fasta.gz -d $WD/c2.index.k31.rtsk -l $WD/plain.txt -o $WD/out2 2>&1 | grep "correction phase" | sed "s/^/plain, ONE reader thread: /"
cmp $WD/out.2.fastq $WD/out2.2.fastq && echo "outputs identical"
( time gzip -1 -k $WD/c2.lr.fq ) 2>&1 | grep real
for i in 1 2 3 4 5 6; do echo $WD/c2.lr.fq.gz >> $WD/gz.txt; done
RTK_CLI_STATS=1 timeout 600 $EXE correct -1 -c 16 -B 64000000 -g $WD/c2.index.k31.fasta.gz -d $WD/c2.index.k31.rtsk -l $WD/gz.txt -o $WD/out3 2>&1 | grep "correction phase" | sed "s/^/gzip input (one inflate stream): /"
cmp $WD/out.2.fastq $WD/out3.2.fastq && echo "gz output identical"
rm -rf $WD

#!/bin/bash
# CLI file to file on ONE gzip member (an ordinary `gzip reads.fastq`): the reader's own inflate (common/finflate.hpp) against zlib inside the same
# reader (RTK_ZLIB_INFLATE=1) and against the one-thread gzread path (-c 1). Developer measurement (DESIGN_HISTORY.md section 5).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WD=$(mktemp -d /tmp/rtk_sgz_XXXX)
python - <<PY
import sys; sys.path.insert(0, ".")
import bench
bench.make_dataset("$WD", 5000000, 150000000, snps=True)
PY
EXE=ratatosk_amd/bin/Ratatosk
for i in 1 2 3 4; do cat $WD/c2.lr.fq >> $WD/plain.fq; done
( time gzip -6 -c $WD/plain.fq > $WD/single.fq.gz ) 2>&1 | grep real
ls -la $WD/single.fq.gz $WD/plain.fq | awk '{print $5, $9}'
( time ratatosk_amd/bin/rtk_gunzip $WD/single.fq.gz --check -@ 2 ) 2>&1 | grep -v "^$" | tr '\n' ' '; echo
( time env RTK_ZLIB_INFLATE=1 ratatosk_amd/bin/rtk_gunzip $WD/single.fq.gz --check -@ 2 ) 2>&1 | grep -v "^$" | tr '\n' ' '; echo
for C in 16; do timeout 120 $EXE correct -1 --parse-only -c $C -B 64000000 -l $WD/single.fq.gz | tail -1 | sed "s/^/own inflate -c $C: /"; RTK_ZLIB_INFLATE=1 timeout 120 $EXE correct -1 --parse-only -c $C -B 64000000 -l $WD/single.fq.gz | tail -1 | sed "s/^/zlib in the reader -c $C: /"; done
RTK_CLI_STATS=1 timeout 300 $EXE correct -1 -c 16 -B 64000000 -g $WD/c2.index.k31.fasta.gz -d $WD/c2.index.k31.rtsk -l $WD/single.fq.gz -o $WD/out 2>&1 | grep "correction phase" | sed "s/^/one gzip member, own inflate, -c 16: /"
RTK_ZLIB_INFLATE=1 RTK_CLI_STATS=1 timeout 300 $EXE correct -1 -c 16 -B 64000000 -g $WD/c2.index.k31.fasta.gz -d $WD/c2.index.k31.rtsk -l $WD/single.fq.gz -o $WD/out1 2>&1 | grep "correction phase" | sed "s/^/one gzip member, zlib on its own thread, -c 16: /"
RTK_CLI_STATS=1 timeout 300 $EXE correct -1 -c 1 -B 64000000 -g $WD/c2.index.k31.fasta.gz -d $WD/c2.index.k31.rtsk -l $WD/single.fq.gz -o $WD/out2 2>&1 | grep "correction phase" | sed "s/^/one gzip member, -c 1 (gzread on the reader thread): /"
cmp $WD/out.2.fastq $WD/out1.2.fastq && cmp $WD/out.2.fastq $WD/out2.2.fastq && echo "outputs identical"
rm -rf $WD

#!/bin/bash
# first-pass CLI at steady state (formatter threads): ONE 7 GB file (the generated long reads 24 times, 3.5 Gb) as plain text, as blocked gzip
# and as an ordinary gzip stream; the reader alone on each (developer; results: profiles/r03_host_io.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WD=$(mktemp -d /tmp/rtk_io_XXXX)
python - <<PY
import sys; sys.path.insert(0, ".")
import bench
bench.make_dataset("$WD", 5000000, 150000000, snps=True)
PY
EXE=ratatosk_amd/bin/Ratatosk
for i in $(seq 24); do cat $WD/c2.lr.fq >> $WD/big.fq; done
ls -la $WD/big.fq | awk '{print "FASTQ bytes:", $5}'
( time ratatosk_amd/bin/rtk_bgzip $WD/big.fq $WD/big.bgzf.gz -@ 32 ) 2>&1 | grep real | sed 's/^/rtk_bgzip -@ 32: /'
ls -la $WD/big.bgzf.gz | awk '{print "BGZF bytes:", $5}'
for i in 1 2 3 4 5 6; do cat $WD/c2.lr.fq; done | gzip -1 > $WD/six.gz
for C in 1 4 16 32; do timeout 120 $EXE correct -1 --parse-only -c $C -B 64000000 -l $WD/big.fq 2>&1 | tail -1 | sed "s/^/plain, parse-only -c $C: /"; done
for C in 1 4 16 32 64; do timeout 120 $EXE correct -1 --parse-only -c $C -B 64000000 -l $WD/big.bgzf.gz 2>&1 | tail -1 | sed "s/^/BGZF, parse-only -c $C: /"; done
timeout 120 $EXE correct -1 --parse-only -c 16 -B 64000000 -l $WD/six.gz 2>&1 | tail -1 | sed "s/^/ordinary gzip (6 copies), parse-only: /"
for V in "3 16 64000000 big.fq" "3 16 64000000 big.fq" "3 16 32000000 big.fq" "3 32 64000000 big.bgzf.gz" "3 32 64000000 big.bgzf.gz" "3 16 64000000 big.bgzf.gz" "3 16 64000000 six.gz"; do set -- $V
  RTK_CLI_STATS=1 timeout 300 $EXE correct -1 -c $2 --workers-per-gpu $1 -B $3 -g $WD/c2.index.k31.fasta.gz -d $WD/c2.index.k31.rtsk -l $WD/$4 -o $WD/out_$4 2>&1 | grep "correction phase" | sed "s/^/$4 workers $1 -c $2 -B $3: /"
done
cmp $WD/out_big.fq.2.fastq $WD/out_big.bgzf.gz.2.fastq && echo "BGZF input: output identical to the plain run"
RTK_CLI_TRACE=1 RTK_CLI_STATS=1 timeout 300 $EXE correct -1 -c 16 -B 64000000 -g $WD/c2.index.k31.fasta.gz -d $WD/c2.index.k31.rtsk -l $WD/big.fq -o $WD/out 2> gpurun_out/cli_trace3.txt; tail -1 gpurun_out/cli_trace3.txt
rm -rf $WD

#!/bin/bash
# CLI file to file on gzip input of several members (cat of gzip files: common/mgzip.hpp) against the same text as ONE member and as a plain file,
# then the CLI tests of the gpu tier on the same build. Developer measurement (DESIGN_HISTORY.md section 5).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WD=$(mktemp -d /tmp/rtk_mgz_XXXX)
python - <<PY
import sys; sys.path.insert(0, ".")
import bench
bench.make_dataset("$WD", 5000000, 150000000, snps=True)
PY
EXE=ratatosk_amd/bin/Ratatosk
split -n l/12 -d $WD/c2.lr.fq $WD/part.        # twelve files of whole lines (records may straddle them: the reader does not care)
ls $WD/part.* | xargs -P 12 -n 1 gzip -1
cat $WD/part.*.gz > $WD/one_copy.gz
for i in 1 2 3 4 5 6 7 8; do cat $WD/one_copy.gz >> $WD/members.fq.gz; cat $WD/c2.lr.fq >> $WD/plain.fq; done   # 96 members
ls -la $WD/members.fq.gz $WD/plain.fq | awk '{print $5, $9}'
for C in 1 8 16 32; do timeout 120 $EXE correct -1 --parse-only -c $C -B 64000000 -l $WD/members.fq.gz | tail -1 | sed "s/^/-c $C: /"; done
for C in 4 16 32; do
RTK_CLI_STATS=1 timeout 300 $EXE correct -1 -c $C -B 64000000 -g $WD/c2.index.k31.fasta.gz -d $WD/c2.index.k31.rtsk -l $WD/members.fq.gz -o $WD/out 2>&1 | grep "correction phase" | sed "s/^/96 gzip members, -c $C: /"
done
RTK_CLI_STATS=1 timeout 300 $EXE correct -1 -c 1 -B 64000000 -g $WD/c2.index.k31.fasta.gz -d $WD/c2.index.k31.rtsk -l $WD/members.fq.gz -o $WD/out1 2>&1 | grep "correction phase" | sed "s/^/96 gzip members, -c 1 (gzread): /"
RTK_CLI_STATS=1 timeout 300 $EXE correct -1 -c 16 -B 64000000 -g $WD/c2.index.k31.fasta.gz -d $WD/c2.index.k31.rtsk -l $WD/plain.fq -o $WD/out2 2>&1 | grep "correction phase" | sed "s/^/plain file, -c 16: /"
cmp $WD/out.2.fastq $WD/out2.2.fastq && cmp $WD/out1.2.fastq $WD/out2.2.fastq && echo "outputs identical"
rm -rf $WD
timeout 120 python -m pytest tests/test_cli.py -m gpu -x -q 2>&1 | tail -2

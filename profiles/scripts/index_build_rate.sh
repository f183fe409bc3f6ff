#!/bin/bash
# index build of the configs[2] set (60 Mb diploid reference, 30x PE150): plain tool, --fast, --gpu; files compared (developer / profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_index_build.py -x -q -m gpu 2>&1 | tail -3
WD=$(mktemp -d /tmp/rtk_idx_XXXX); B=ratatosk_amd/bin
REF=${1:-60000000}
( time $B/rtk_simulate --prefix $WD/c2 --seed 3 --ref-len $REF --het 0.001 --sr-cov 30 --sr-err 0.005 --lr-cov 0.05 --lr-len 8000 --lr-profile ont --lr-err 0.07 ) 2>&1 | grep real
ls -la $WD/c2.sr.fq | awk '{print "short-read FASTQ bytes:", $5}'
for mode in "--gpu" "--fast" ""; do
  echo "== rtk_build_index $mode"
  ( time RTK_INDEX_TRACE=1 timeout 1500 $B/rtk_build_index -s $WD/c2.sr.fq -o $WD/out$mode --snps $mode ) 2>&1 | grep "\[\|real\|meet\|solid\|unitigs$"
done
cmp $WD/out.index.k31.fasta.gz $WD/out--gpu.index.k31.fasta.gz && cmp $WD/out.index.k31.rtsk $WD/out--gpu.index.k31.rtsk && echo "gpu files identical to the plain tool's"
cmp $WD/out.index.k31.fasta.gz $WD/out--fast.index.k31.fasta.gz && cmp $WD/out.index.k31.rtsk $WD/out--fast.index.k31.rtsk && echo "fast files identical to the plain tool's"
rm -rf $WD

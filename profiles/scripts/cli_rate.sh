#!/bin/bash
# end-to-end rate of the stand-alone CLI (parse FASTQ, correct on one GPU with two overlapped workers, write FASTQ in input order)
set -e
W=$(mktemp -d)
ratatosk_amd/bin/rtk_simulate --prefix $W/c2 --seed 2 --ref-len 5000000 --sr-cov 30 --sr-err 0.005 --lr-cov 30 --lr-len 8000 --lr-profile ont --lr-err 0.07 2>/dev/null
ratatosk_amd/bin/rtk_build_index -s $W/c2.sr.fq -o $W/c2 2>/dev/null
BASES=$(awk 'NR%4==2{n+=length($0)}END{print n}' $W/c2.lr.fq)
for c in 1; do
  S=$(date +%s.%N)
  ratatosk_amd/bin/Ratatosk correct -1 -v -c $c -g $W/c2.index.k31.fasta.gz -d $W/c2.index.k31.rtsk -l $W/c2.lr.fq -o $W/out | grep -v Processed
  E=$(date +%s.%N)
  python3 -c "print('CLI -c $c: %d bases in %.2f s = %.1f M bases/s (graph load included)' % ($BASES, $E-$S, $BASES/($E-$S)/1e6))"
done
ls -la $W/out.2.fastq | awk '{print "output bytes", $5}'
rm -rf $W

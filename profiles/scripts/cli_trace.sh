#!/bin/bash
# CLI correction-phase wall time for several ticket sizes (-B)
set -e
W=$(mktemp -d)
ratatosk_amd/bin/rtk_simulate --prefix $W/c2 --seed 2 --ref-len 5000000 --sr-cov 30 --sr-err 0.005 --lr-cov 30 --lr-len 8000 --lr-profile ont --lr-err 0.07 2>/dev/null
ratatosk_amd/bin/rtk_build_index -s $W/c2.sr.fq -o $W/c2 2>/dev/null
for B in 16000000 32000000 67108864; do
  echo "-B $B"
  RTK_TRACE=1 ratatosk_amd/bin/Ratatosk correct -1 -v -c 1 -B $B -g $W/c2.index.k31.fasta.gz -d $W/c2.index.k31.rtsk -l $W/c2.lr.fq -o $W/out 2>&1 | grep -E "phase|correct_batch|seeds attempt|regions done"
done
rm -rf $W

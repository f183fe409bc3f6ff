#!/bin/bash
# second pass file to file, A/B of environment settings on the same files (round 5). Every setting runs twice, a pause between runs (the memory of the process before is
# still being given back when the next one reserves its work areas: without the pause every other run starts short of memory and is slow). Prints the rate of the
# correction phase and the seconds of graph load + reservations.       usage: r05_pass2_ab.sh "A=1,B=2;C=3" [REF=60e6] [LR=128e6] [COPIES=18]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_pass2; mkdir -p $OUT
timeout 1500 python profiles/scripts/pass2_rate.py ${2:-60e6} ${3:-128e6} 63 > $OUT/p2rate_ab.json 2> $OUT/p2rate_ab.err
WD=$(ls -d /tmp/rtk_p2_* | tail -1)
rm -f $WD/in.txt $WD/raw.txt; for i in $(seq ${4:-18}); do echo $WD/c2.2.fastq >> $WD/in.txt; echo $WD/c2.lr.fq >> $WD/raw.txt; done
run() { sleep 4; echo "== $*"; env "$@" RTK_CLI_STATS=1 timeout 600 ratatosk_amd/bin/Ratatosk correct -2 -K 63 -c 16 -g $WD/c2.p2.index.k63.fasta.gz -d $WD/c2.p2.index.k63.rtsk -l $WD/in.txt -L $WD/raw.txt -o $WD/again 2>&1 | grep "correction phase" | sed 's/thread-seconds.*//'; }
run A=0 > /dev/null
IFS=';' read -ra SPECS <<< "${1:-A=0}"
for rep in $(seq ${REPS:-2}); do for s in "${SPECS[@]}"; do run $(echo $s | tr ',' ' '); done; done
RTK_TRACE=1 timeout 300 ratatosk_amd/bin/Ratatosk correct -2 -K 63 -c 16 --workers-per-gpu 1 -g $WD/c2.p2.index.k63.fasta.gz -d $WD/c2.p2.index.k63.rtsk -l $WD/c2.2.fastq -L $WD/c2.lr.fq -o $WD/again 2>&1 | grep "phase attempt\|alignment skipped\|k_phase_long wave-0" | head -9

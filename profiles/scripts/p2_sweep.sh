#!/bin/bash
# second pass at steady state: waves per workgroup of the long class, grid sizes, tickets in flight (developer)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python profiles/scripts/pass2_rate.py 5e6 128e6 63 > /tmp/p2rate.json 2>/tmp/p2rate.err
WD=$(ls -d /tmp/rtk_p2_* | tail -1)
for i in $(seq ${COPIES:-18}); do echo $WD/c2.2.fastq >> $WD/in.txt; echo $WD/c2.lr.fq >> $WD/raw.txt; done
run() { env "$@" RTK_CLI_STATS=1 timeout 300 ratatosk_amd/bin/Ratatosk correct -2 -K 63 -c 16 -g $WD/c2.p2.index.k63.fasta.gz -d $WD/c2.p2.index.k63.rtsk -l $WD/in.txt -L $WD/raw.txt -o $WD/again 2>&1 | grep "correction phase" | sed "s/^.*correction phase/$* : /; s/thread-seconds.*//"; }
run A=0
run A=0
run RTK_PHASE_PGRID=2048 RTK_PHASE_MGRID=1600
run RTK_PHASE_PGRID=2048 RTK_PHASE_MGRID=1600
run RTK_PHASE_MGRID=1600
run RTK_PHASE_PGRID=1536 RTK_PHASE_MGRID=1152
run A=0
run RTK_P2_RGRID=1024
run RTK_P2_RGRID=256

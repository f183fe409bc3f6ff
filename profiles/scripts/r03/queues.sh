cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 python profiles/scripts/pass2_rate.py 5e6 128e6 63 > /tmp/p2rate.json 2>/tmp/p2rate.err
WD=$(ls -d /tmp/rtk_p2_* | tail -1)
for i in 1 2 3 4 5 6; do echo $WD/c2.2.fastq >> $WD/in.txt; echo $WD/c2.lr.fq >> $WD/raw.txt; done
for W in "" "--workers-per-gpu 6"; do
RTK_CLI_STATS=1 timeout 120 ratatosk_amd/bin/Ratatosk correct -2 -K 63 -c 16 $W -g $WD/c2.p2.index.k63.fasta.gz -d $WD/c2.p2.index.k63.rtsk -l $WD/in.txt -L $WD/raw.txt -o $WD/again 2>&1 | grep "correction phase"
done

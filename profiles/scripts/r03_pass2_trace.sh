cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RTK_PHASE_LONG=24576 RTK_PHASE_LGRID=64 RTK_PHASE_MGRID=128
timeout 1200 python profiles/scripts/pass2_rate.py 5e6 256e6 63 --workers-per-gpu 5 > gpurun_out/pass2_t.json 2> gpurun_out/pass2_t.err; echo "rc=$?"
WD=$(ls -d /tmp/rtk_p2_* | tail -1)
RTK_TRACE=1 RTK_CLI_STATS=1 ratatosk_amd/bin/Ratatosk correct -2 -K 63 -c 16 --workers-per-gpu 5 -g $WD/c2.p2.index.k63.fasta.gz -d $WD/c2.p2.index.k63.rtsk -l $WD/c2.2.fastq -L $WD/c2.lr.fq -o $WD/again > /dev/null 2> gpurun_out/pass2_trace_full.txt
grep -v "size class\|fine shares\|shares of\|cycle shares" gpurun_out/pass2_trace_full.txt | head -150

#!/bin/bash
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/rtk_wd; O=gpurun_out/r06_split_ab.txt; : > $O
for rep in 1 2; do for sp in 2 3 4; do
  RTK_COALESCE_SPLIT=$sp timeout 500 python profiles/scripts/r06/tickets_quick.py /tmp/rtk_wd "groups = around/$sp" 2>/dev/null >> $O
done; done
# the CLI by ticket size
PRE=/tmp/rtk_wd/c2
for i in $(seq 26); do echo $PRE.lr.fq; done > /tmp/rtk_wd/cli_in.txt
for B in 33554432 67108864 100663296; do for rep in 1 2; do
  RTK_CLI_STATS=1 timeout 300 ratatosk_amd/bin/Ratatosk correct -1 -c 16 --gpus 1 -g $PRE.index.k31.fasta.gz -d $PRE.index.k31.rtsk -l /tmp/rtk_wd/cli_in.txt -o /tmp/rtk_wd/cli_out -B $B 2>&1 | grep -o "correction phase [0-9.]* s wall, [0-9]* bases" | awk -v b=$B '{printf "CLI -1 -B %s: %s s  %.4g bases/s\n", b, $3, $6/$3}' >> $O; rm -f /tmp/rtk_wd/cli_out.2.fastq
done; done
cat $O

#!/bin/bash
# configs[4] as BASELINE.json states it, in one call on one GPU box: 3 Gb diploid reference, 30x short reads sampled inside the index tool, the graph resident in HBM;
# 10x long reads (30 Gb: the generated 0.4 Gb set given to `Ratatosk correct -1` as many times as it takes, one -l each) FILE TO FILE through the shipped driver;
# then per-kernel roofline figures and rocprofv3 passes (kernel stats; FETCH_SIZE, WRITE_SIZE each on its own) over tickets on the same resident graph.
#   usage: r05_config4.sh [REF_MB=3000] [SR_COV=30] [LR_GB=30]
set -u
export TMPDIR=/tmp RTK_C4_DIR=/tmp/c4
OUT=$PWD/gpurun_out/r05_config4${RTK_C4_TAG:-}; mkdir -p $OUT $RTK_C4_DIR   # RTK_C4_SNPS=1: index with --snps; RTK_C4_REAL_OUT=1: the corrected FASTQ written to the disk when it has room
REF_MB=${1:-3000}; SR_COV=${2:-30}; LR_GB=${3:-30}
( time RTK_C4_KEEP=1 RTK_C4_INDEX_LOG=$OUT/build_index.log RTK_C4_INDEX_TIMEOUT=${RTK_C4_INDEX_TIMEOUT:-1500} RTK_C4_OUT=$OUT/r05_config4_run.json timeout 3000 python profiles/scripts/config4_run.py $REF_MB $SR_COV 128 6 > $OUT/c4_full.log 2>&1 ) 2> $OUT/c4_full_time.txt
tail -3 $OUT/c4_full.log | cut -c1-400; tail -3 $OUT/c4_full_time.txt
PRE=/tmp/c4/c4_keep/c4
[ -f $PRE.index.k31.rtsk ] || exit 1
# ---- file to file: LR_GB of long reads through the driver (the output goes to /dev/null through a symlink: 2 x LR_GB of FASTQ would not fit the box's disk) ----
LRB=$(awk 'NR%4==2{n+=length($0)}END{print n}' $PRE.lr.fq); REPS=$(python -c "print(max(1, round($LR_GB*1e9/$LRB)))")
ARGS=""; for i in $(seq 1 $REPS); do ARGS="$ARGS -l $PRE.lr.fq"; done
rm -f /tmp/c4/f2f.2.fastq; FREE_GB=$(df --output=avail -BG /tmp/c4 | tail -1 | tr -dc 0-9); NEED_GB=$(python -c "print(int(2.1*$LR_GB)+8)")
if [ "${RTK_C4_REAL_OUT:-0}" = 1 ] && [ "$FREE_GB" -gt "$NEED_GB" ]; then echo "output to the disk ($FREE_GB GB free, $NEED_GB needed)" > $OUT/cli_out.txt; else ln -s /dev/null /tmp/c4/f2f.2.fastq; echo "output to /dev/null ($FREE_GB GB free, $NEED_GB needed)" > $OUT/cli_out.txt; fi
( time RTK_CLI_STATS=1 timeout 1500 ratatosk_amd/bin/Ratatosk correct -1 -c 16 --gpus 1 -g $PRE.index.k31.fasta.gz -d $PRE.index.k31.rtsk $ARGS -o /tmp/c4/f2f > $OUT/cli.log 2>&1 ) 2> $OUT/cli_time.txt
echo "cli rc $? reps $REPS bases_per_file $LRB" >> $OUT/cli.log; ls -laL /tmp/c4/f2f.2.fastq >> $OUT/cli_out.txt; [ -L /tmp/c4/f2f.2.fastq ] || { head -c 2000000000 /tmp/c4/f2f.2.fastq | awk 'NR%4==2{n+=length($0)}END{print "bases in the first 2 GB of the output:", n}' >> $OUT/cli_out.txt; rm -f /tmp/c4/f2f.2.fastq; }
cat $OUT/cli_out.txt; tail -4 $OUT/cli.log | cut -c1-600
# ---- per-kernel roofline at this scale + rocprofv3 passes ----
P=$OUT/prof; mkdir -p $P
STEPS="python profiles/scripts/config4_steps.py 3 128"
RTK_C4_ROOFLINE_OUT=$OUT/r05_config4_roofline.json timeout 600 $STEPS > $P/steps.log 2>&1; tail -1 $P/steps.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_serial -o stats -- $STEPS > $P/stats.log 2> $P/stats.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch -o fetch -- $STEPS > /dev/null 2> $P/fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write -o write -- $STEPS > /dev/null 2> $P/write.err
mkdir -p $P/stats $P/calib_fetch $P/calib_write $P/sq
python profiles/scripts/summarise.py $P $OUT r05_config4 > $P/summarise.log 2>&1
find $P -name "*.csv" -size +8M -delete   # (traces of thousands of launches: only the summaries travel back)
ls -la $OUT | head -30

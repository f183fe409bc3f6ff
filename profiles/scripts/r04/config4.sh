#!/bin/bash
# configs[4] for real, in one call on one GPU box: the 3 Gb x 30x run (index kept), then rocprofv3 passes over tickets on the resident 3 Gb graph
# (kernel trace + stats; FETCH_SIZE, WRITE_SIZE and SQ counters each in a pass of its own). Every step under a timeout of its own; the index tool's log is
# written as it goes (gpurun_out/r04_config4/build_index.log).      usage: r04_config4.sh [REF_MB=3000] [SR_COV=30] [TICKETS=6]
set -u
export TMPDIR=/tmp RTK_C4_DIR=/tmp/c4
OUT=$PWD/gpurun_out/r04_config4; mkdir -p $OUT $RTK_C4_DIR
REF_MB=${1:-3000}; SR_COV=${2:-30}; TICKETS=${3:-6}
( time RTK_C4_KEEP=1 RTK_C4_INDEX_LOG=$OUT/build_index.log RTK_C4_INDEX_TIMEOUT=${RTK_C4_INDEX_TIMEOUT:-1200} RTK_C4_OUT=$OUT/r04_config4_dry_run.json timeout 2700 python profiles/scripts/r04_config4.py $REF_MB $SR_COV 128 $TICKETS > $OUT/c4_full.log 2>&1 ) 2> $OUT/c4_full_time.txt
tail -3 $OUT/c4_full.log | cut -c1-600; tail -3 $OUT/c4_full_time.txt; tail -30 $OUT/build_index.log
[ -f /tmp/c4/c4_keep/c4.index.k31.rtsk ] || exit 1
P=$OUT/prof; mkdir -p $P
STEPS="python profiles/scripts/r04_config4_steps.py 3 128"
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_serial -o stats -- $STEPS > $P/stats.log 2> $P/stats.err
timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch -o fetch -- $STEPS > /dev/null 2> $P/fetch.err
timeout 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write -o write -- $STEPS > /dev/null 2> $P/write.err
timeout 420 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $P/sq -o sq -- $STEPS > /dev/null 2> $P/sq.err
mkdir -p $P/stats $P/calib_fetch $P/calib_write
python profiles/scripts/summarise.py $P $OUT r04_config4 > $P/summarise.log 2>&1
cat $P/stats.log | tail -2; ls -la $OUT

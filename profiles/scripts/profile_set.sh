#!/bin/bash
# Profile set of a round (usage: profile_set.sh rNN) (run from the repo root on the GPU box through gpurun): rocprofv3 kernel trace + stats of the default command, the PMC passes
# (one counter group per pass, steps back to back), the same counter passes on configs[1], counter calibration, then the bench line itself.
set -u
R=${1:-r05}
OUT=$PWD/gpurun_out/prof_$R
SUM=$PWD/gpurun_out/profiles_$R
WD=/tmp/rtk_${R}_wd
mkdir -p $OUT $SUM $WD
export TMPDIR=/tmp
BENCH="python bench.py --workdir $WD --no-cpu-baseline --no-host-legs --no-config1-leg"
SERIAL="python bench.py --workdir $WD --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs --no-config1-leg --serial"
$SERIAL > /dev/null 2> $OUT/warm.err   # builds the data set once
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $SUM/${R}_bench_under_rocprof.json 2> $OUT/stats.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -o stats -- $BENCH --serial > /dev/null 2> $OUT/stats_serial.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $SERIAL > /dev/null 2> $OUT/fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $SERIAL > /dev/null 2> $OUT/write.err
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/sq -o sq -- $SERIAL > /dev/null 2> $OUT/sq.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -o calib -- python profiles/scripts/calib_gather.py > /dev/null 2> $OUT/calib_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calib_write -o calib -- python profiles/scripts/calib_gather.py > /dev/null 2> $OUT/calib_write.err
python profiles/scripts/summarise.py $OUT $SUM $R > $OUT/summarise.log 2>&1
# the same counters on configs[1] (the set the round-3 review's targets are written on)
OUT1=$PWD/gpurun_out/prof_${R}_c1; mkdir -p $OUT1
C1="python bench.py --workdir $WD --config1-only --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs --serial"
$C1 > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT1/stats_serial -o stats -- $C1 > /dev/null 2> $OUT1/stats.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT1/fetch -o fetch -- $C1 > /dev/null 2> $OUT1/fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT1/write -o write -- $C1 > /dev/null 2> $OUT1/write.err
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT1/sq -o sq -- $C1 > /dev/null 2> $OUT1/sq.err
mkdir -p $OUT1/calib_fetch $OUT1/calib_write
python profiles/scripts/summarise.py $OUT1 $SUM ${R}_config1 > $OUT1/summarise.log 2>&1
# the bench line itself (with the host legs, the configs[1] leg and the CPU baseline), after the counter summary it quotes
cp $SUM/${R}_pmc_summary.json profiles/${R}_pmc_summary.json
( time timeout 1500 python bench.py --workdir $WD > $SUM/${R}_bench.json 2> $OUT/bench.err ) 2> $SUM/${R}_bench_time.txt
ls -la $SUM; tail -3 $SUM/${R}_bench_time.txt; tail -c 600 $SUM/${R}_bench.json

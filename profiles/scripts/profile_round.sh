#!/bin/bash
# Usage: bash profiles/scripts/profile_round.sh rNN   (run from the repo root on the GPU box, through gpurun)
# Writes raw rocprofv3 output under gpurun_out/prof_<round>/ and the summaries to be committed under gpurun_out/profiles_<round>/.
set -u
R=${1:-r01}
OUT=$PWD/gpurun_out/prof_$R
SUM=$PWD/gpurun_out/profiles_$R
mkdir -p $OUT $SUM
export TMPDIR=/tmp
BENCH="python bench.py --no-cpu-baseline --no-host-legs"          # the default command (steps overlap on two streams)
SERIAL="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs --serial"   # counter passes: one kernel at a time, so that a dispatch's counters are its own
# 2. kernel trace + stats of the same command
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $SUM/${R}_bench_under_rocprof.json 2> $OUT/stats.err
# 2b. the same with the steps back to back (per-kernel durations undisturbed by the next step's seed kernels)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -o stats -- python bench.py --no-cpu-baseline --no-host-legs --serial > /dev/null 2> $OUT/stats_serial.err
# 3. PMC passes, each on its own (FETCH_SIZE and WRITE_SIZE do not fit one pass)
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $SERIAL > /dev/null 2> $OUT/fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $SERIAL > /dev/null 2> $OUT/write.err
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/sq -o sq -- $SERIAL > /dev/null 2> $OUT/sq.err
# 4. counter calibration on known byte counts
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -o calib -- python profiles/scripts/calib_gather.py > /dev/null 2> $OUT/calib_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calib_write -o calib -- python profiles/scripts/calib_gather.py > /dev/null 2> $OUT/calib_write.err
python profiles/scripts/summarise.py $OUT $SUM $R
# 5. the bench line itself (with the host legs and the CPU baseline), after the counter summary it quotes: roofline.traffic is read from profiles/<round>_pmc_summary.json
cp $SUM/${R}_pmc_summary.json profiles/${R}_pmc_summary.json
timeout 900 python bench.py > $SUM/${R}_bench.json 2> $OUT/bench.err
ls -la $SUM

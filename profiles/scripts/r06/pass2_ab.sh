#!/bin/bash
# Second pass, file to file, by ticket size (-B) and tickets in flight (--workers-per-gpu): the steady-state list-file run of bench.py's second_pass leg on the 60 Mb set.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/rtk_wd; O=gpurun_out/r06_pass2_ab.txt; : > $O
PRE=$(python - <<'PY'
import bench
print(bench.make_dataset("/tmp/rtk_wd", 60_000_000, int(4.3 * 64_000_000) + 200_000, snps=True, het=0.001))
PY
)
EXE=ratatosk_amd/bin/Ratatosk; OUT=/tmp/rtk_wd/p2_out
$EXE correct -1 -c 16 --gpus 1 -g $PRE.index.k31.fasta.gz -d $PRE.index.k31.rtsk -l $PRE.lr.fq -o $OUT > /dev/null 2>&1
[ -f $OUT.p2.index.k63.rtsk ] || ratatosk_amd/bin/rtk_build_index -s $PRE.sr.fq --colour-reads $OUT.2.fastq -k 63 -o $OUT.p2 2> /dev/null
for i in $(seq 18); do echo $OUT.2.fastq; done > $OUT.p2in.txt; for i in $(seq 18); do echo $PRE.lr.fq; done > $OUT.p2raw.txt
run() { RTK_CLI_STATS=1 timeout 300 $EXE correct -2 -c 16 --gpus 1 -g $OUT.p2.index.k63.fasta.gz -d $OUT.p2.index.k63.rtsk -l $OUT.p2in.txt -L $OUT.p2raw.txt -o $OUT "$@" 2>&1 | grep -o "correction phase [0-9.]* s wall, [0-9]* bases" | awk -v a="$*" '{printf "%-40s %s s  %.3g bases/s\n", a, $3, $6/$3}' >> $O; rm -f $OUT.fastq; }
run; run
for B in 16777216 67108864 134217728; do run -B $B; done
for W in 4 6 10 12; do run --workers-per-gpu $W; done
run -B 67108864 --workers-per-gpu 6
run
cat $O

#!/bin/bash
# Second pass: why do identical runs give 5.3 ... 8.9 x 10^8 bases/s? Five default runs; the library's own notes of new device / pinned memory taken INSIDE the run (RTK_TRACE) and CLI stats.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/rtk_wd; O=gpurun_out/r06_pass2_trace.txt; : > $O
PRE=$(python - <<'PY'
import bench
print(bench.make_dataset("/tmp/rtk_wd", 60_000_000, int(4.3 * 64_000_000) + 200_000, snps=True, het=0.001))
PY
)
EXE=ratatosk_amd/bin/Ratatosk; OUT=/tmp/rtk_wd/p2_out
[ -f $OUT.2.fastq ] || $EXE correct -1 -c 16 --gpus 1 -g $PRE.index.k31.fasta.gz -d $PRE.index.k31.rtsk -l $PRE.lr.fq -o $OUT > /dev/null 2>&1
[ -f $OUT.p2.index.k63.rtsk ] || ratatosk_amd/bin/rtk_build_index -s $PRE.sr.fq --colour-reads $OUT.2.fastq -k 63 -o $OUT.p2 2> /dev/null
for i in $(seq 18); do echo $OUT.2.fastq; done > $OUT.p2in.txt; for i in $(seq 18); do echo $PRE.lr.fq; done > $OUT.p2raw.txt
for rep in 1 2 3 4 5; do
  RTK_CLI_STATS=1 RTK_TRACE=1 timeout 300 $EXE correct -2 -c 16 --gpus 1 -g $OUT.p2.index.k63.fasta.gz -d $OUT.p2.index.k63.rtsk -l $OUT.p2in.txt -L $OUT.p2raw.txt -o $OUT "$@" > gpurun_out/p2trace_$rep.log 2>&1; rm -f $OUT.fastq
  echo "== run $rep: $(grep -o 'correction phase [0-9.]* s wall, [0-9]* bases' gpurun_out/p2trace_$rep.log | awk '{printf "%s s  %.3g bases/s", $3, $6/$3}')" >> $O
  echo "   new device memory: $(grep -c 'new device memory' gpurun_out/p2trace_$rep.log) calls, $(grep 'new device memory' gpurun_out/p2trace_$rep.log | awk '{for(i=1;i<=NF;i++) if($i=="in") s+=$(i+1)} END{printf "%.1f ms", s}'); new pinned: $(grep -c 'new pinned' gpurun_out/p2trace_$rep.log) calls, $(grep 'new pinned' gpurun_out/p2trace_$rep.log | awk '{for(i=1;i<=NF;i++) if($i=="in") s+=$(i+1)} END{printf "%.1f ms", s}'); slow fetches (>5 ms to the host): $(grep 'MB to the host' gpurun_out/p2trace_$rep.log | awk '{if ($(NF-1)+0 > 5) c++} END{print c+0}') of $(grep -c 'MB to the host' gpurun_out/p2trace_$rep.log)" >> $O
  grep -E "tickets in flight|workers|thread-seconds|stats" gpurun_out/p2trace_$rep.log | head -4 | cut -c1-300 >> $O
done
grep -E "phase_take|reserve" gpurun_out/p2trace_1.log | head -10 | cut -c1-200 >> $O
cat $O

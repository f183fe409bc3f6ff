#!/bin/bash
# instruction-cache / scalar-cache / L2 counters of k_regions (one --pmc pass per group, steps back to back)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_ic; rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|TCC_[A-Z_]*REQ[A-Z_a-z]*\|TCC_HIT[a-z_]*\|TCC_MISS[a-z_]*\|SQ_INSTS_[A-Z_]*" | sort -u | tr '\n' ' ' > $OUT/avail.txt
SERIAL="python bench.py --config1-only --steps 2 --warmup 1 --no-cpu-baseline --no-host-legs --serial"
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU"; do
  i=$((i+1)); timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p -- $SERIAL > /dev/null 2> $OUT/p$i.err
done
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_ic/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_regions"):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-24s per launch %.4g  (%d launches)" % (k, tot[k] / max(1, n[k]), n[k]))
PY
tail -3 $OUT/p1.err; cat $OUT/avail.txt | head -c 3000

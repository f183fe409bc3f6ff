#!/bin/bash
# Developer run: HBM byte counters and vector-memory instruction counts of k_regions (one --pmc pass per group, steps back to back).
# Raw counter units as rocprofv3 reports them; profiles/scripts/summarise.py holds the calibration used for the committed summaries.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcq; rm -rf $OUT; mkdir -p $OUT
SERIAL="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-legs --serial"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p -- $SERIAL > /dev/null 2> $OUT/p$i.err
done
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("gpurun_out/pmcq/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_regions"):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-24s per launch %.4g  (%d launches)" % (k, tot[k] / max(1, n[k]), n[k]))
PY

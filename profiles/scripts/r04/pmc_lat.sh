#!/bin/bash
# average latencies of k_regions' memory instructions (level counters / instruction counts)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_lat; rm -rf $OUT; mkdir -p $OUT
SERIAL="python bench.py --config1-only --steps 2 --warmup 1 --no-cpu-baseline --no-host-legs --serial"
i=0
for grp in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_WAVE_CYCLES" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM" "SQ_IFETCH_LEVEL SQ_IFETCH SQ_WAIT_INST_LDS SQ_WAIT_ANY" "TCC_READ_REQ_LATENCY_sum TCC_READ_REQ_sum" "TCC_WRITE_REQ_LATENCY_sum TCC_WRITE_REQ_sum" "TCC_PERF_SEL_EA_RDREQ_sum TCC_PERF_SEL_EA_RDREQ_LEVEL_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_LEVEL_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_LEVEL_sum"; do
  i=$((i+1)); timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p -- $SERIAL > /dev/null 2> $OUT/p$i.err
done
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_lat/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_regions"):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-32s per launch %.4g  (%d launches)" % (k, tot[k] / max(1, n[k]), n[k]))
PY
grep -l -i "error\|invalid\|not found" $OUT/*.err | head

#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_sqc; rm -rf $OUT; mkdir -p $OUT
SERIAL="python bench.py --config1-only --steps 2 --warmup 1 --no-cpu-baseline --no-host-legs --serial"
i=0
for grp in "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_STALL SQC_TC_REQ" "SQC_DCACHE_BUSY_CYCLES SQC_DCACHE_INPUT_VALID_READYB SQC_TC_DATA_READ_REQ SQC_TC_DATA_WRITE_REQ" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_BUSY_CU_CYCLES SQ_WAVES SQ_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" "TCC_EA_RDREQ_sum TCC_EA_RD_UNCACHED_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum"; do
  i=$((i+1)); timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p -- $SERIAL > /dev/null 2> $OUT/p$i.err
done
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_sqc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_regions"):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-32s per launch %.4g  (%d launches)" % (k, tot[k] / max(1, n[k]), n[k]))
PY

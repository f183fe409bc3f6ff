#!/bin/bash
# Short form of r04_tail.sh: wave-level figures of k_regions in a traced run (both sets) + the TA / TCP counters of configs[1] that say whether the
# vector-memory path of the CUs is what the kernel's constant term is.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_tail; rm -rf $OUT; mkdir -p $OUT
SERIAL="python bench.py --config1-only --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs"
RTK_TRACE=1 timeout 300 $SERIAL 2> $OUT/trace_config1.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config1', round(d['value']/1e9,4), d['roofline']['kernel_ms_per_step'])"
grep "k_regions waves\|k_regions attempt\|by size class\|k_regions shares\|fine shares" $OUT/trace_config1.txt | tail -6
RTK_TRACE=1 timeout 600 python bench.py --no-config1-leg --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs 2> $OUT/trace_60mb.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('60Mb', round(d['value']/1e9,4), d['roofline']['kernel_ms_per_step'])"
grep "k_regions waves\|k_regions attempt\|by size class\|k_regions shares\|fine shares" $OUT/trace_60mb.txt | tail -6
i=0
for grp in "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1)); timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p -- $SERIAL > /dev/null 2> $OUT/p$i.err
done
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("gpurun_out/r04_tail/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_regions(") or r["Kernel_Name"] == "k_regions":
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-44s per launch %.4g  (%d launches)" % (k, tot[k] / max(1, n[k]), n[k]))
PY
grep -l -i "error\|invalid\|not found" $OUT/*.err | head

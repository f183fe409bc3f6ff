#!/bin/bash
# Where k_regions' time goes at the level of whole waves (traced run: first start to last end, average lifetime, the longest region) and whether the
# vector-memory path of the CUs (texture addresser / L1 tag lookups) is what saturates: TA / TCP counters per launch.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_tail; rm -rf $OUT; mkdir -p $OUT
SERIAL="python bench.py --config1-only --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs"
RTK_TRACE=1 timeout 300 $SERIAL 2> $OUT/trace_config1.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config1', round(d['value']/1e9,4), d['roofline']['kernel_ms_per_step'])"
grep "k_regions waves\|k_regions attempt" $OUT/trace_config1.txt | tail -4
RTK_TRACE=1 timeout 600 python bench.py --no-config1-leg --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs 2> $OUT/trace_60mb.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('60Mb', round(d['value']/1e9,4), d['roofline']['kernel_ms_per_step'])"
grep "k_regions waves\|k_regions attempt" $OUT/trace_60mb.txt | tail -4
rocprofv3 --list-avail 2>/dev/null | grep -o "\b\(TCP\|TA\|TD\)_[A-Z0-9_a-z]*" | sort -u > $OUT/avail_tcp_ta.txt; wc -l $OUT/avail_tcp_ta.txt
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_ATOMIC_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1)); timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p -- $SERIAL > /dev/null 2> $OUT/p$i.err
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

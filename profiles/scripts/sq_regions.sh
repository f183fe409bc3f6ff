#!/bin/bash
# SQ counters of the region kernel (serial steps so that nothing else shares the CUs)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/sq_now; mkdir -p $OUT
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/a -o sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --serial > /dev/null 2> $OUT/a.err
timeout 900 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/b -o sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --serial > /dev/null 2> $OUT/b.err
python - <<'PY'
import csv, glob, collections
for d in ("a","b"):
    f = glob.glob("gpurun_out/sq_now/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f: print("no csv for", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in ("k_regions", "k_inexact", "k_finalize", "k_mask"):
        if k in acc: print(k, {c: "%.3g" % (sum(v)/len(v)) for c, v in acc[k].items()})
PY

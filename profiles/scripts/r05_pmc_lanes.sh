#!/bin/bash
# Developer run: SQ counters of k_regions_lanes / k_regions on one resident 64 Mb batch of configs[1] (one --pmc pass per group).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcl; rm -rf $OUT; mkdir -p $OUT
CMD="python profiles/scripts/r05_lanes_ab.py c1 64000000 ${1:-256}"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p -- $CMD > /dev/null 2> $OUT/p$i.err
done
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("gpurun_out/pmcl/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0]
        if kn.startswith("k_regions"):
            tot[(kn, r["Counter_Name"])] += float(r["Counter_Value"]); n[(kn, r["Counter_Name"])] += 1
for k in sorted(tot): print("%-18s %-24s per launch %.4g  (%d launches)" % (k[0], k[1], tot[k] / max(1, n[k]), n[k]))
PY

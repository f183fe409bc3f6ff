#!/bin/bash
# Round 6, GPU call 7: waves of a half-area region kernel (seed kernels of the next group need slots), 1 Mi x 16 and 4 Mi x 8 callers; then configs[4] with the untimed first group
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/rtk_wd; O=gpurun_out/r06_call7.txt; : > $O
for hw in 2048 1536 1024; do for rep in 1 2; do
RTK_HALF_WAVES=$hw timeout 600 python - >> $O 2>/dev/null <<PY
import os, sys
sys.path.insert(0, os.getcwd()); os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import bench
from ratatosk_amd import api
pre = bench.make_dataset("/tmp/rtk_wd", 60_000_000, int(4.3 * 64_000_000) + 200_000, snps=True, het=0.001)
g = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, n_threads=64)
seqs, quals = bench.read_long_reads(pre + ".lr.fq", 140_000_000)
res = []
for mib, callers, n_t in ((1, 16, 48), (1, 8, 48), (4, 8, 32), (4, 16, 32)):
    want = mib << 20
    tickets, cs, cq, cur = [], [], [], 0
    for s_, q_ in zip(seqs, quals):
        cs.append(s_); cq.append(q_); cur += len(s_)
        if cur >= want:
            tickets.append((cs, cq)); cs, cq, cur = [], [], 0
    best = max((bench.correct_batch_leg(api, g, g.opts(), tickets, n_t, callers) for _ in range(3)), key=lambda r: r.get("value", 0))
    res.append("%dMi x %d: %.3g (%.1f per launch)" % (mib, callers, best["value"], best["tickets"] / max(1, best["launch_groups"])))
print("RTK_HALF_WAVES=$hw", "; ".join(res))
PY
done; done
( time timeout 1500 python bench_config4.py gpurun_out/r06_config4.json 3000 30 16 128 /tmp/rtk_c4 > gpurun_out/r06_config4.log 2>&1 ) 2>> $O
python -c "
import json; d=json.load(open('gpurun_out/r06_config4.json')); print(d.get('tickets'), d.get('first_group_s_with_allocations'), d.get('index'), d.get('skipped'))" >> $O
cat $O

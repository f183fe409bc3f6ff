"""bases/s through rtk_correct_batch at the ticket sizes / caller counts the round-5 review names (1 Mi x 16, 1 Mi x 8, 4 Mi x 8, 4 Mi x 16), best of three, on the bench's 60 Mb set.
Usage: python profiles/scripts/r06/tickets_quick.py [workdir] [label]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import bench
from ratatosk_amd import api
wd = sys.argv[1] if len(sys.argv) > 1 else "/tmp/rtk_wd"
pre = bench.make_dataset(wd, 60_000_000, int(4.3 * 64_000_000) + 200_000, snps=True, het=0.001)
g = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, n_threads=64)
seqs, quals = bench.read_long_reads(pre + ".lr.fq", 140_000_000)
res = []
for mib, callers, n_t in ((1, 16, 48), (1, 8, 48), (4, 8, 32), (4, 16, 32), (1, 3, 24), (4, 3, 16)):
    want = mib << 20
    tickets, cs, cq, cur = [], [], [], 0
    for s_, q_ in zip(seqs, quals):
        cs.append(s_); cq.append(q_); cur += len(s_)
        if cur >= want:
            tickets.append((cs, cq)); cs, cq, cur = [], [], 0
    best = max((bench.correct_batch_leg(api, g, g.opts(), tickets, n_t, callers) for _ in range(3)), key=lambda r: r.get("value", 0))
    res.append("%dMi x %d: %.3g (%.1f per launch)" % (mib, callers, best["value"], best["tickets"] / max(1, best["launch_groups"])))
print(sys.argv[2] if len(sys.argv) > 2 else "", "; ".join(res))

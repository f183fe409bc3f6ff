"""One traced run of the ticket coalescing: 16 callers x 1 Mi tickets (and 8 x 4 Mi) through rtk_correct_batch with RTK_TRACE=1: where a group's time goes."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import bench
from ratatosk_amd import api
wd = sys.argv[1] if len(sys.argv) > 1 else "/tmp/rtk_wd"
pre = bench.make_dataset(wd, 60_000_000, int(4.3 * 64_000_000) + 200_000, snps=True, het=0.001)
g = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, n_threads=64)
seqs, quals = bench.read_long_reads(pre + ".lr.fq", 140_000_000)
for mib, callers, n_t in ((1, 16, 48), (4, 8, 32)):
    want = mib << 20
    tickets, cs, cq, cur = [], [], [], 0
    for s_, q_ in zip(seqs, quals):
        cs.append(s_); cq.append(q_); cur += len(s_)
        if cur >= want:
            tickets.append((cs, cq)); cs, cq, cur = [], [], 0
    r = bench.correct_batch_leg(api, g, g.opts(), tickets, n_t, callers)  # warm
    os.environ["RTK_TRACE"] = "1"
    sys.stderr.write("==== %d Mi x %d callers\n" % (mib, callers)); sys.stderr.flush()
    r = bench.correct_batch_leg(api, g, g.opts(), tickets, n_t, callers)
    os.environ.pop("RTK_TRACE")
    sys.stderr.write("==== result %s\n" % r); sys.stderr.flush()

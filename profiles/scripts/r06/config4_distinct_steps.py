"""Loads the configs[4] index bench_config4.py left under WORKDIR and runs N DIFFERENT 64 Mb tickets once each, one at a time: the command the rocprofv3 passes of r06/config4_pmc.sh wrap
(kernel stats; FETCH_SIZE and WRITE_SIZE each in a pass of its own). Prints the per-kernel HIP-event times. Usage: config4_distinct_steps.py WORKDIR [N=4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from ratatosk_amd import api
import bench
wd = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pre = os.path.join(wd, "c4")
g = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, n_threads=128)
seqs, quals = bench.read_long_reads(pre + ".lr.fq", (n + 1) * 64_000_000)
tickets, cs, cq, cur = [], [], [], 0
for s_, q_ in zip(seqs, quals):
    cs.append(s_); cq.append(q_); cur += len(s_)
    if cur >= 64_000_000:
        tickets.append((cs, cq)); cs, cq, cur = [], [], 0
b = api.Batch(g, *tickets[0]); b.run(g.opts()); b.close()  # (the first ticket allocates the work areas and the batch buffers: its launches are the first of every kernel in the trace)
for t in tickets[1:n + 1]:
    b = api.Batch(g, *t); b.run(g.opts()); st = b.stats(); b.close()
    print("ticket of %d bases:" % st["in_bases"], {k: round(v, 2) for k, v in st.items() if k.startswith("ms_")})

"""Loads the index r04_config4.py left under $RTK_C4_DIR/c4_keep and runs N tickets (one at a time): the command the rocprofv3 passes of
r04_config4.sh wrap. Usage: python profiles/scripts/r04_config4_steps.py [N=3] [THREADS=128]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from ratatosk_amd import api
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 128
pre = os.path.join(os.environ.get("RTK_C4_DIR", "/tmp"), "c4_keep", "c4")
g = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, n_threads=threads)
seqs, quals = bench.read_long_reads(pre + ".lr.fq", 64_000_000)
b = api.Batch(g, seqs, quals)
for i in range(n + 1):
    b.run(g.opts())
print("ran", n + 1, "tickets of", b.in_bases, "bases;", {k: round(v, 2) for k, v in b.stats().items() if k.startswith("ms_")})

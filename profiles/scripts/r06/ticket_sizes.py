"""bases/s through rtk_correct_batch by ticket size and number of callers (bench.py's by_ticket_size leg on its own, on the bench's 60 Mb set: a workdir that holds it is reused).
Usage: python profiles/scripts/r06/ticket_sizes.py [workdir]     RTK_COALESCE_BASES=0 in the environment: no merging (round-5 behaviour)"""
import json, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import bench
from ratatosk_amd import api
wd = sys.argv[1] if len(sys.argv) > 1 else "/tmp/rtk_wd"
os.makedirs(wd, exist_ok=True)
pre = bench.make_dataset(wd, 60_000_000, int(4.3 * 64_000_000) + 200_000, snps=True, het=0.001)
g = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, n_threads=64)
seqs, quals = bench.read_long_reads(pre + ".lr.fq", 140_000_000)
out = bench.ticket_size_leg(types.SimpleNamespace(), api, g, g.opts(), [(seqs, quals)])
print(json.dumps(out))
for k, v in out.items():
    if isinstance(v, dict):
        print(k, " ".join("%s=%.3g" % (c, v[c]) for c in v if "callers" in c and isinstance(v[c], (int, float))))

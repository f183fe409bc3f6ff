"""Wall-clock per stage call vs HIP-event kernel sums (host-overhead check). Usage: python profiles/scripts/stage_timing.py"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ratatosk_amd import api
api.load_library(None)
wd = tempfile.mkdtemp(prefix="rtk_st_")
pre = bench.make_dataset(wd, 5_000_000, 48_000_000)
g = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, device=0)
seqs, quals = bench.read_long_reads(pre + ".lr.fq", 32_000_000)
half = len(seqs) // 2
bs = [api.Batch(g, seqs[:half], quals[:half]), api.Batch(g, seqs[half:], quals[half:])]
o = g.opts()
for rep in range(3):
    for i, b in enumerate(bs):
        t0 = time.time(); b.run_seeds(o); t1 = time.time(); b.run_regions(o); t2 = time.time()
        st = b.stats()
        print("rep %d batch %d: seeds %.1f ms (kernels %.1f)  regions %.1f ms (kernels %.1f)" % (rep, i, 1e3 * (t1 - t0), st["ms_lookup_exact"] + st["ms_mask"] + st["ms_lookup_inexact"] + st["ms_seeds"],
              1e3 * (t2 - t1), st["ms_regions"] + st["ms_correct"] + st["ms_stitch"]))
for rep in range(2):
    t0 = time.time(); api.run_pipelined(bs + bs, o); t1 = time.time()
    print("pipelined 4 steps: %.1f ms/step" % (1e3 * (t1 - t0) / 4))

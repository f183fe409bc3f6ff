"""How many reads of configs[1] depend on the reading of assumption [A2] (`or_exclusive_match` of the 1-edit k-mer search, src/Graph.cpp:193)?
The same 32 Mb of long reads corrected twice on the device, rtk_opts.a2_exclusive = 0 (union of the three edit kinds, the default) and 1
(first kind that matches, substitution > insertion > deletion). Usage (GPU box): python profiles/scripts/a2_count.py"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ratatosk_amd import api
wd = tempfile.mkdtemp(prefix="rtk_a2_")
pre = bench.make_dataset(wd, 5_000_000, 40_000_000, snps=True)
g = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31)
seqs, quals = bench.read_long_reads(pre + ".lr.fq", 32_000_000)
res = []
for x in (0, 1):
    o = g.opts(); o.a2_exclusive = x
    b = api.Batch(g, seqs, quals); b.run(o); res.append(b.fetch()); st = b.stats()
    print("a2_exclusive=%d: raw 1-edit hits %d, regions %d" % (x, st["n_hits_inexact"], st["n_regions"]), file=sys.stderr)
diff = sum(1 for a, b_ in zip(res[0], res[1]) if a[0] != b_[0])
dq = sum(1 for a, b_ in zip(res[0], res[1]) if a[0] == b_[0] and a[1] != b_[1])
bases = sum(len(s) for s in seqs)
import difflib
ed = 0
print(json.dumps({"workload": "configs[1]: 5 Mb reference, 30x short reads, ONT-profile long reads, SNP-annotated index", "reads": len(seqs), "bases": bases,
                  "reads_with_different_sequence": diff, "reads_with_same_sequence_but_different_qualities": dq}))

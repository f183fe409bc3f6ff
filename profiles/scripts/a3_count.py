"""How many reads depend on the reading of assumption [A3] (order of getSuccessors() on the reverse strand, which breaks ties between candidate
paths of equal score in exploreSubGraph)? 32 Mb of long reads corrected twice on the device, rtk_opts.a3_strand_order = 0 and 1, on configs[1]
(haploid 5 Mb reference) and on a diploid 5 Mb reference with 0.1 % heterozygous SNPs (the bubbles of configs[2]). Usage (GPU box): python profiles/scripts/a3_count.py"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ratatosk_amd import api
out = {}
for tag, het in (("configs[1] (haploid)", 0.0), ("diploid, 0.1 % heterozygous SNPs", 0.001)):
    wd = tempfile.mkdtemp(prefix="rtk_a3_")
    pre = bench.make_dataset(wd, 5_000_000, 40_000_000, snps=True, het=het)
    g = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31)
    seqs, quals = bench.read_long_reads(pre + ".lr.fq", 32_000_000)
    res = []
    for x in (0, 1):
        o = g.opts(); o.a3_strand_order = x
        b = api.Batch(g, seqs, quals); b.run(o); res.append(b.fetch())
    out[tag] = {"reads": len(seqs), "bases": sum(len(s) for s in seqs), "reads_with_different_sequence": sum(1 for a, b_ in zip(res[0], res[1]) if a[0] != b_[0]),
                "reads_with_same_sequence_but_different_qualities": sum(1 for a, b_ in zip(res[0], res[1]) if a[0] == b_[0] and a[1] != b_[1]),
                "bases_that_differ_where_lengths_agree": sum(sum(1 for x_, y_ in zip(a[0], b_[0]) if x_ != y_) for a, b_ in zip(res[0], res[1]) if len(a[0]) == len(b_[0]))}
print(json.dumps(out))

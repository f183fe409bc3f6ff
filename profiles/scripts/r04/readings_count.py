"""How many reads do the unverifiable choices decide? 32 Mb of long reads corrected on the device under every reading of
  [A2] or_exclusive_match (rtk_opts.a2_exclusive = 1 exclusive sub->ins->del [default], 2 exclusive ins->del->sub, 0 union),
  [A3] order of getSuccessors() on the reverse strand (a3_strand_order = 0 walk [default], 1 strand),
  [D1] tie order of chooseColors' anchors (d1_desc = 0 ascending unitig id [default], 1 descending),
each against the default, on configs[1] (haploid 5 Mb reference) and on a diploid 5 Mb reference with 0.1 % heterozygous SNPs (the bubbles of the
chr20-scale set). Writes profiles/r04_{a2,a3,d1}_count.json when run from the repo root on a GPU box: python profiles/scripts/r04_readings_count.py OUTDIR"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ratatosk_amd import api
outdir = sys.argv[1] if len(sys.argv) > 1 else "."
res = {"a2": {}, "a3": {}, "d1": {}}
for tag, het in (("configs[1] (haploid)", 0.0), ("diploid, 0.1 % heterozygous SNPs", 0.001)):
    wd = tempfile.mkdtemp(prefix="rtk_rd_")
    pre = bench.make_dataset(wd, 5_000_000, 40_000_000, snps=True, het=het)
    g = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31)
    seqs, quals = bench.read_long_reads(pre + ".lr.fq", 32_000_000)
    def run(**kw):
        o = g.opts()
        for k_, v in kw.items():
            setattr(o, k_, v)
        b = api.Batch(g, seqs, quals); b.run(o); r = b.fetch(); st = b.stats(); b.close()
        return r, st
    base, st0 = run(a2_exclusive=1, a3_strand_order=0, d1_desc=0)
    def diff(other):
        return {"reads": len(seqs), "bases": sum(len(s) for s in seqs),
                "reads_with_different_sequence": sum(1 for a, b_ in zip(base, other) if a[0] != b_[0]),
                "reads_with_same_sequence_but_different_qualities": sum(1 for a, b_ in zip(base, other) if a[0] == b_[0] and a[1] != b_[1]),
                "bases_that_differ_where_lengths_agree": sum(sum(1 for x_, y_ in zip(a[0], b_[0]) if x_ != y_) for a, b_ in zip(base, other) if len(a[0]) == len(b_[0]))}
    for name, kw in (("exclusive-ids (ins->del->sub) vs exclusive (sub->ins->del, default)", dict(a2_exclusive=2)), ("union vs exclusive (default)", dict(a2_exclusive=0))):
        r, st = run(**kw); d = diff(r); d["raw_1edit_hits"] = st["n_hits_inexact"]; d["raw_1edit_hits_default"] = st0["n_hits_inexact"]; res["a2"].setdefault(tag, {})[name] = d
    r, _ = run(a3_strand_order=1); res["a3"][tag] = diff(r)
    r, _ = run(d1_desc=1); res["d1"][tag] = diff(r)
for key, what in (("a2", "[A2] readings of or_exclusive_match, each against the default (exclusive, substitution -> insertion -> deletion)"), ("a3", "[A3] strand order of getSuccessors() against the default (walk order)"), ("d1", "[D1] descending against ascending (default) unitig id among anchors of equal colour-set cardinality in chooseColors (src/Correction.cpp:286-293)")):
    json.dump({"what": what, "command": "python profiles/scripts/r04_readings_count.py", "sets": res[key]}, open(os.path.join(outdir, "r04_%s_count.json" % key), "w"), indent=1)
print(json.dumps(res))

"""A/B of the lane-per-region kernel on one resident batch: kernel times (HIP events inside the library) by RTK_LANE_MAX_GAP / RTK_LANE_WAVES,
and a digest of the corrected batch for every setting (the lane kernel and the wave kernel must write the same bytes).
usage: python profiles/scripts/r05_lanes_ab.py [c1|c2] [batch_bases] [settings ...]   setting = max_gap[:waves]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from oracle import oracle_py as op
from ratatosk_amd import api

which = sys.argv[1] if len(sys.argv) > 1 else "c1"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64_000_000
settings = sys.argv[3:] or ["0", "64", "128", "256"]
ref_len, het = (5_000_000, 0.0) if which == "c1" else (60_000_000, 0.001)
work = os.environ.get("RTK_BENCH_WORK", "/tmp/rtk_bench")
os.makedirs(work, exist_ok=True)
pre = bench.make_dataset(work, ref_len, batch, snps=True, het=het, fast="--gpu", name=which)
fa, rt = pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk"
pg = api.Graph(fa, rt, 31, device=0)
reads = op.read_fastq(pre + ".lr.fq")
seqs, quals, tot = [], [], 0
for r in reads:
    if tot >= batch: break
    seqs.append(r[1]); quals.append(r[2]); tot += len(r[1])
print("batch: %d reads, %d bases" % (len(seqs), tot), flush=True)
ref_digest = None
for s in settings:
    gap, _, waves = s.partition(":")
    os.environ["RTK_LANE_MAX_GAP"] = gap
    if waves: os.environ["RTK_LANE_WAVES"] = waves
    else: os.environ.pop("RTK_LANE_WAVES", None)
    b = api.Batch(pg, seqs, quals)
    best = None
    for rep in range(3):
        b.run(pg.opts()); st = b.stats()
        if best is None or st["ms_total"] < best["ms_total"]: best = st
    got = b.fetch()
    h = hashlib.sha256()
    for g_ in got: h.update(g_[0].encode()); h.update(g_[1].encode())
    d = h.hexdigest()[:16]
    if ref_digest is None: ref_digest = d
    print("gap<%s waves=%s: total %.2f ms | correct %.2f (lanes %.2f, wave kernel %.2f) | seeds %.2f | lane regions %d handed %d (%.1f%%) of %d regions | digest %s %s" % (
        gap, waves or "default", best["ms_total"], best["ms_correct"], best["ms_lanes"], best["ms_correct"] - best["ms_lanes"],
        best["ms_lookup_exact"] + best["ms_mask"] + best["ms_lookup_inexact"] + best["ms_seeds"], best["n_lane_regions"], best["n_lane_handed"],
        100.0 * best["n_lane_handed"] / max(1, best["n_lane_regions"]), best["n_regions"], d, "OK" if d == ref_digest else "DIFFERS"), flush=True)
    del b

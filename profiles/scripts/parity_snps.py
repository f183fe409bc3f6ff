"""SNP-annotated index at volume (outside the test tiers): a diploid graph indexed with `rtk_build_index --snps`, N Mb of ONT-profile
long reads through the HIP path vs the oracle (all host threads), and the kernel times on the annotated vs the plain index."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ratatosk_amd import api
from oracle import oracle_py as op
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ref_len = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
het = float(sys.argv[3]) if len(sys.argv) > 3 else 0.002
api.load_library(None)
wd = tempfile.mkdtemp(prefix="rtk_snps_")
bin_dir = os.path.join(ROOT, "ratatosk_amd", "bin"); pre = os.path.join(wd, "c")
subprocess.check_call([os.path.join(bin_dir, "rtk_simulate"), "--prefix", pre, "--seed", "9", "--ref-len", str(ref_len), "--het", str(het), "--sr-cov", "30", "--sr-err", "0.005",
                       "--lr-cov", "%.3f" % ((mb + 1) * 1e6 / ref_len), "--lr-len", "8000", "--lr-profile", "ont", "--lr-err", "0.07"], stderr=subprocess.DEVNULL)
t0 = time.time()
r = subprocess.run([os.path.join(bin_dir, "rtk_build_index"), "-s", pre + ".sr.fq", "-o", pre, "--snps"], capture_output=True, text=True)
print([l for l in r.stderr.splitlines() if "SNP" in l][0], "(%.1f s)" % (time.time() - t0))
subprocess.check_call([os.path.join(bin_dir, "rtk_build_index"), "-s", pre + ".sr.fq", "-o", pre + "_plain"], stderr=subprocess.DEVNULL)
reads = op.read_fastq(pre + ".lr.fq")
seqs, quals, tot = [], [], 0
for _, s, q in reads:
    seqs.append(s); quals.append(q); tot += len(s)
    if tot >= mb * 1_000_000:
        break
res = {}
for name, p in (("annotated", pre), ("plain", pre + "_plain")):
    g = api.Graph(p + ".index.k31.fasta.gz", p + ".index.k31.rtsk", 31, device=0)
    b = api.Batch(g, seqs, quals)
    b.run(); b.run()
    st = b.stats()
    res[name] = b.fetch()
    print("%-9s ms: seeds %.2f regions %.2f total %.2f" % (name, st["ms_seeds"], st["ms_correct"], st["ms_total"]))
og = op.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31)
t1 = time.time(); want, _ = og.correct_batch(seqs, quals, threads=os.cpu_count() or 8); t2 = time.time()
bad = [i for i, (a, b) in enumerate(zip(res["annotated"], want)) if a != b]
changed = sum(1 for a, b in zip(res["annotated"], res["plain"]) if a != b)
print("SNP volume parity: %d reads, %d bases, oracle %.1f s, reads changed by the annotations: %d, mismatching reads: %d" % (len(seqs), tot, t2 - t1, changed, len(bad)))
sys.exit(1 if bad else 0)

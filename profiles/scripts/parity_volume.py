"""Volume parity check outside the test tiers: N Mb of configs[1] long reads through the HIP path vs the oracle (all host threads)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ratatosk_amd import api
from oracle import oracle_py as op
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ref_len = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000     # 60000000 + het 0.001 = configs[2]'s graph
het = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
api.load_library(None)
wd = tempfile.mkdtemp(prefix="rtk_pv_")
if het > 0.0:
    import subprocess
    bin_dir = os.path.join(ROOT, "ratatosk_amd", "bin"); pre = os.path.join(wd, "c")
    subprocess.check_call([os.path.join(bin_dir, "rtk_simulate"), "--prefix", pre, "--seed", "3", "--ref-len", str(ref_len), "--het", str(het), "--sr-cov", "30", "--sr-err", "0.005",
                           "--lr-cov", "%.3f" % ((mb + 20) * 1e6 / ref_len), "--lr-len", "8000", "--lr-profile", "ont", "--lr-err", "0.07"], stderr=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(bin_dir, "rtk_build_index"), "-s", pre + ".sr.fq", "-o", pre], stderr=subprocess.DEVNULL)
else:
    pre = bench.make_dataset(wd, ref_len, 150_000_000)
fa, rt = pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk"
g, og = api.Graph(fa, rt, 31, device=0), op.Graph(fa, rt, 31)
seqs, quals = bench.read_long_reads(pre + ".lr.fq", 150_000_000)
# skip the first 16 Mb (bench.py already checks those on every run)
skip, tot, i0 = 16_000_000, 0, 0
while tot < skip:
    tot += len(seqs[i0]); i0 += 1
sel_s, sel_q, tot = [], [], 0
for s, q in zip(seqs[i0:], quals[i0:]):
    sel_s.append(s); sel_q.append(q); tot += len(s)
    if tot >= mb * 1_000_000:
        break
t0 = time.time(); got = g.correct_batch(sel_s, sel_q); t1 = time.time()
want, _ = og.correct_batch(sel_s, sel_q, threads=os.cpu_count() or 8); t2 = time.time()
bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
print("volume parity: %d reads, %d bases, GPU %.1f s (host-inclusive), oracle %.1f s, mismatching reads: %d" % (len(sel_s), tot, t1 - t0, t2 - t1, len(bad)))
sys.exit(1 if bad else 0)

"""Volume parity of the second pass beside the test tiers (round 5: the whole-read alignment of phasing() is skipped or pruned on the device, the oracle aligns everything):
MB megabases of pass-1 reads (corrected by the oracle, so that the input is the oracle's own) + the raw reads through `correct -2` on the device and through the oracle, k2 = 63.
Usage: python profiles/scripts/r05_pass2_volume_parity.py [MB=32] [REF=5000000]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ratatosk_amd import api
from oracle import oracle_py as op
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ref = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
wd = tempfile.mkdtemp(prefix="rtk_p2v_")
pre = bench.make_dataset(wd, ref, (mb + 4) * 1_000_000, fast="" if os.environ.get("RTK_LIB_OVERRIDE") else "--gpu")  # (RTK_LIB_OVERRIDE = the host simulator: a dry run of this script without a GPU)
thr = len(os.sched_getaffinity(0))
og1 = op.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31)
seqs, quals = bench.read_long_reads(pre + ".lr.fq", mb * 1_000_000)
names = ["r%d" % i for i in range(len(seqs))]
t0 = time.time(); out1, _ = og1.correct_batch(seqs, quals, threads=thr); t1 = time.time()
p1 = pre + ".pass1.fq"
with open(p1, "w") as f:
    for n_, (s_, q_) in zip(names, out1):
        f.write("@%s\n%s\n+\n%s\n" % (n_, s_, q_))
subprocess.check_call([os.path.join(ROOT, "ratatosk_amd", "bin", "rtk_build_index"), "-s", pre + ".sr.fq", "--colour-reads", p1, "-k", "63", "-o", pre + ".p2"], stderr=subprocess.DEVNULL)
fa, rt = pre + ".p2.index.k63.fasta.gz", pre + ".p2.index.k63.rtsk"
og, pg = op.Graph(fa, rt, 63), api.Graph(fa, rt, 63, device=0)
s1, q1 = [o[0] for o in out1], [o[1] for o in out1]
t2 = time.time(); b = api.Batch(pg, s1, q1, raw=seqs); b.run(pg.opts(long_read_correct=1)); got = b.fetch(); st = b.stats(); t3 = time.time()
want = og.correct_batch2(s1, q1, seqs, og.opts(long_read_correct=1), threads=thr); t4 = time.time()
bad = [i for i, (a, c) in enumerate(zip(got, want)) if a != c]
changed = sum(1 for (s_, _), s0 in zip(want, s1) if s_ != s0)
print("second-pass volume parity: %d reads, %d bases (longest %d), k2 = 63; whole-read alignment skipped for %d reads, pruned for the others; device %.1f s, oracle pass 1 %.1f s + pass 2 %.1f s on %d threads; reads the pass changes: %d; mismatching reads: %d"
      % (len(seqs), sum(len(s) for s in seqs), max(len(s) for s in seqs), st["n_phase_skipped"], t3 - t2, t1 - t0, t4 - t3, thr, changed, len(bad)))
sys.exit(1 if bad else 0)

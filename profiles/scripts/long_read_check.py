"""One-off check: reads longer than the 128 kb base size of the per-wave slabs (scratch regrowth path) equal the oracle."""
import os, sys, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ratatosk_amd import api
from oracle import oracle_py as op
api.load_library(None)
wd = tempfile.mkdtemp(prefix="rtk_lr_"); pre = os.path.join(wd, "c"); bin_dir = os.path.join(ROOT, "ratatosk_amd", "bin")
subprocess.check_call([os.path.join(bin_dir, "rtk_simulate"), "--prefix", pre, "--seed", "9", "--ref-len", "600000", "--het", "0.001", "--sr-cov", "30", "--sr-err", "0.005",
                       "--lr-n", "4", "--lr-len", "200000", "--lr-profile", "uniform", "--lr-err", "0.07"], stderr=subprocess.DEVNULL)
subprocess.check_call([os.path.join(bin_dir, "rtk_build_index"), "-s", pre + ".sr.fq", "-o", pre], stderr=subprocess.DEVNULL)
fa, rt = pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk"
reads = op.read_fastq(pre + ".lr.fq")
print("read lengths", [len(r[1]) for r in reads])
g, og = api.Graph(fa, rt, 31, device=0), op.Graph(fa, rt, 31)
got = g.correct_batch([r[1] for r in reads], [r[2] for r in reads])
want, _ = og.correct_batch([r[1] for r in reads], [r[2] for r in reads], threads=8)
print("long reads identical:", got == want)
sys.exit(0 if got == want else 1)

"""Freezes BASELINE.json configs[0] ("tiny synthetic: 50 kb random ref, 1k x 150 bp PE short reads, 100 x 10 kb 10 %-error long reads,
`Ratatosk correct -1`") as checksums in tests/golden/config0.json:

  inputs   sha256 of the generated long reads, of the unitig FASTA (decompressed) and of the .rtsk written by the repo's own
           seeded generator + index producer -- so that a change of the generators is told apart from a change of the correction;
  output   sha256 of the corrected FASTQ (`@name\\nseq\\n+\\nqual\\n` records, input order: src/Ratatosk.cpp:518-519,919) as the
           ORACLE produced it when this file was frozen, plus per-read CRC32s of (seq, qual) to localise a difference.

There is no reference binary (Bifrost absent) and the reference ships no fixtures, so this pins the repo against ITSELF over time:
oracle and HIP path must both keep reproducing these bytes (tests/test_configs.py). Run by hand after a deliberate change:
    python tests/golden/gen_config0_golden.py
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

SIM_ARGS = ["--seed", "1", "--ref-len", "50000", "--sr-pairs", "1000", "--sr-len", "150", "--sr-err", "0.001",
            "--lr-n", "100", "--lr-len", "10000", "--lr-err", "0.10"]


def make(workdir):
    bin_dir = os.path.join(ROOT, "ratatosk_amd", "bin")
    pre = os.path.join(workdir, "c0")
    subprocess.check_call([os.path.join(bin_dir, "rtk_simulate"), "--prefix", pre] + SIM_ARGS, stderr=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(bin_dir, "rtk_build_index"), "-s", pre + ".sr.fq", "-o", pre], stderr=subprocess.DEVNULL)
    return pre


def input_sums(pre):
    sha = lambda b: hashlib.sha256(b).hexdigest()
    return {"lr.fq": sha(open(pre + ".lr.fq", "rb").read()), "index.fasta": sha(gzip.open(pre + ".index.k31.fasta.gz", "rb").read()),
            "index.rtsk": sha(open(pre + ".index.k31.rtsk", "rb").read())}


def fastq_bytes(names, recs):
    return "".join("@%s\n%s\n+\n%s\n" % (n, s, q) for n, (s, q) in zip(names, recs)).encode()


def main():
    from oracle import oracle_py as op
    with tempfile.TemporaryDirectory() as d:
        pre = make(d)
        reads = op.read_fastq(pre + ".lr.fq")
        og = op.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31)
        out = {"sim_args": SIM_ARGS, "inputs": input_sums(pre), "n_reads": len(reads), "in_bases": sum(len(r[1]) for r in reads)}
        # the default reading of assumption [A2] (exclusive, since round 4) at the top level; the union reading (the default of rounds 1-3: these
        # are the very bytes frozen then) next to it, so that a change of the default is told apart from a change of the correction
        for key, env in (("", None), ("a2_union", "union")):
            if env is None:
                os.environ.pop("RTK_A2_XOR", None)
            else:
                os.environ["RTK_A2_XOR"] = env
            want, _ = og.correct_batch([r[1] for r in reads], [r[2] for r in reads], threads=os.cpu_count() or 4)
            rec = {"out_bases": sum(len(w[0]) for w in want), "fastq_sha256": hashlib.sha256(fastq_bytes([r[0] for r in reads], want)).hexdigest(),
                   "read_crc32": [zlib.crc32((w[0] + "\n" + w[1]).encode()) for w in want]}
            if key:
                out[key] = rec
            else:
                out.update(rec)
        os.environ.pop("RTK_A2_XOR", None)
    with open(os.path.join(HERE, "config0.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote config0.json:", out["fastq_sha256"], out["a2_union"]["fastq_sha256"])


if __name__ == "__main__":
    main()

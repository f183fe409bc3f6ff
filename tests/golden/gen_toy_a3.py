"""The [A3] toy: a read on which the ORDER of getSuccessors() decides the corrected base, with the expected base derived by hand for both readings
(tests/golden/toy_a3/NOTE.md). Writes tests/golden/toy_a3/{hap.fa, sr.fq, lr.fq}; there is no frozen output: the test computes the expected base
from the rule in the note and from the strand on which the index stores the two flank unitigs.

Genome (k = 31): hapA = 1200 random bases with `A` at position 600, hapB = the same with `C` there. Short reads: error-free 2x100 bp pairs, insert
400, one pair every 5 bp, the two haplotypes ALTERNATING pair by pair (pair 2i from hapA, 2i + 1 from hapB at the same start), so that the lowest
pair ids -- the ones chooseColors takes first -- come from both haplotypes and both branches of the bubble pass the colour filter.
Long reads: hapA[300:900) with `G` at the SNP and hapA[250:850) with `T` there (neither allele: both branches of the bubble align equally well).

Run by hand: python tests/golden/gen_toy_a3.py"""
import os
import random
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "toy_a3")
K, SNP, L = 31, 600, 1200


def rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def haplotypes():
    rnd = random.Random(20260929)
    base = "".join(rnd.choice("ACGT") for _ in range(L))
    return base[:SNP] + "A" + base[SNP + 1:], base[:SNP] + "C" + base[SNP + 1:]


def short_reads(hap_a, hap_b):
    out, pid = [], 0
    for start in range(0, L - 400 + 1, 5):
        for hap in (hap_a, hap_b):
            frag = hap[start:start + 400]
            if (start // 5) % 2:
                frag = rc(frag)
            out.append(("sr%d" % pid, frag[:100])); out.append(("sr%d" % pid, rc(frag)[:100])); pid += 1
    return out


def long_reads(hap_a):
    """(name, read, position of the SNP in the read): haplotype substrings with a base at the SNP that is neither allele"""
    out = []
    for name, lo, hi, third in (("third_base_G_at_the_snp", 300, 900, "G"), ("third_base_T_at_the_snp", 250, 850, "T")):
        t = hap_a[lo:hi]
        out.append((name, t[:SNP - lo] + third + t[SNP - lo + 1:], SNP - lo))
    return out


def write(d=OUT):
    hap_a, hap_b = haplotypes()
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "hap.fa"), "w").write(">hapA\n%s\n>hapB\n%s\n" % (hap_a, hap_b))
    with open(os.path.join(d, "sr.fq"), "w") as f:
        for n, s in short_reads(hap_a, hap_b):
            f.write("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)))
    with open(os.path.join(d, "lr.fq"), "w") as f:
        for n, s, _ in long_reads(hap_a):
            f.write("@%s\n%s\n+\n%s\n" % (n, s, "5" * len(s)))


def build_index(workdir):
    pre = os.path.join(workdir, "toy_a3")
    subprocess.check_call([os.path.join(ROOT, "ratatosk_amd", "bin", "rtk_build_index"), "-s", os.path.join(OUT, "sr.fq"), "-o", pre], stderr=subprocess.DEVNULL)
    return pre


if __name__ == "__main__":
    write()
    print("wrote", OUT)

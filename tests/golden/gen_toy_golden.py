"""Builds the hand-checkable toy fixture of SURVEY.md 8(c): tests/golden/toy/{hap.fa, sr.fq, lr.fq, expected.fastq}.

Genome (k = 31): two 2 kb haplotypes that differ by ONE SNP (position 700: hapA 'A' / hapB 'C' after forcing) and by ONE copy of a
12 bp tandem unit (hapA 8 copies at 1300, hapB 7). Short reads: error-free 2x100 bp pairs, insert 400, one pair every 5 bp on both
haplotypes and both strands alternating -> the graph is exactly the de Bruijn graph of the two haplotypes (every k-mer seen >= 2x).
Long reads: thirteen hand-made reads, each built from a haplotype substring by the explicit edits listed in READS below; what each one
must come out as, and why, is worked out in tests/golden/toy/NOTE.md. expected.fastq is the oracle's output, frozen after the hand
check; tests/test_toy_golden.py holds oracle AND HIP path to it and re-derives the hand-checkable parts independently.

Run by hand: python tests/golden/gen_toy_golden.py [--freeze]   (--freeze rewrites expected.fastq from the oracle)
"""
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
TOY = os.path.join(HERE, "toy")
K = 31
SNP_POS = 700
TANDEM_POS, UNIT, COPIES = 1300, "ACGGTCATTGCA", 8


def rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def haplotypes():
    rnd = random.Random(20260928)
    base = [rnd.choice("ACGT") for _ in range(2000)]
    span = UNIT * COPIES
    base[TANDEM_POS:TANDEM_POS + len(span)] = list(span)
    # keep the flanks of the tandem block from continuing the period by chance
    base[TANDEM_POS - 1] = "T" if UNIT[-1] != "T" else "G"
    base[TANDEM_POS + len(span)] = "T" if UNIT[0] != "T" else "G"
    hap_a = "".join(base)
    hap_a = hap_a[:SNP_POS] + "A" + hap_a[SNP_POS + 1:]
    hap_b = hap_a[:SNP_POS] + "C" + hap_a[SNP_POS + 1:]
    hap_b = hap_b[:TANDEM_POS] + hap_b[TANDEM_POS + len(UNIT):]  # one unit less
    return hap_a, hap_b


def short_reads(hap_a, hap_b):
    out, pid = [], 0
    for h, hap in enumerate((hap_a, hap_b)):
        for start in range(0, len(hap) - 400 + 1, 5):
            frag = hap[start:start + 400]
            if (start // 5) % 2:
                frag = rc(frag)
            m1, m2 = frag[:100], rc(frag)[:100]
            out.append(("sr%d" % pid, m1)); out.append(("sr%d" % pid, m2)); pid += 1
    return out


def edit(s, ops):
    """ops: list of (pos, kind, arg) on the ORIGINAL coordinates, applied right to left. kind: 'sub' (arg = new base), 'del' (arg = n),
    'ins' (arg = string inserted before pos)."""
    for pos, kind, arg in sorted(ops, reverse=True):
        if kind == "sub":
            assert s[pos] != arg
            s = s[:pos] + arg + s[pos + 1:]
        elif kind == "del":
            s = s[:pos] + s[pos + arg:]
        else:
            s = s[:pos] + arg + s[pos:]
    return s


def other(c):
    return {"A": "C", "C": "G", "G": "T", "T": "A"}[c]


def long_reads(hap_a, hap_b):
    rnd = random.Random(7)
    junk = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))
    R = []
    # name, source haplotype substring (truth), the read itself
    t = hap_a[100:900]; R.append(("r0_all_solid", t, t))
    j = junk(500); R.append(("r1_no_solid", j, j))
    t = hap_a[50:650]; R.append(("r2_same_unitig_sub", t, edit(t, [(300, "sub", other(t[300]))])))
    t = hap_a[300:1100]; R.append(("r3_bubble_hapA_del2", t, edit(t, [(380, "del", 2)])))  # 20 bp before the SNP (read pos 400)
    t = hap_b[300:1100]; R.append(("r4_bubble_hapB_ins1", t, edit(t, [(415, "ins", "G")])))  # 15 bp after the SNP
    t = hap_a[200:800]; R.append(("r5_head_errors", t, edit(t, [(6, "sub", other(t[6])), (20, "del", 1)])))
    t = hap_a[200:800]; R.append(("r6_tail_errors", t, edit(t, [(575, "sub", other(t[575])), (590, "ins", "T")])))
    t = hap_a[0:150] + junk(1100) + hap_a[150:300]; R.append(("r7_region_too_long", t, t))
    t = rc(hap_a[900:1700]); R.append(("r8_revcomp_tandem_hapA", t, edit(t, [(100, "sub", other(t[100])), (650, "del", 1)])))
    t = hap_b[1000:1800]; R.append(("r9_tandem_hapB_two_errors", t, edit(t, [(250, "sub", other(t[250])), (500, "ins", "A")])))
    t = hap_a[300:1100]; R.append(("r10_error_cluster_over_snp", t, edit(t, [(385, "sub", other(t[385])), (405, "sub", other(t[405])), (425, "del", 1), (445, "sub", other(t[445]))])))
    # r11 / r12: an interior region with NO exact k-mer for 557 windows (errors 25 bp apart from read 70 = hapA 450 to read 595 = hapA 975): wider than
    # insert_sz = 500, so the stretch is searched for 1-edit k-mers (Graph.cpp:112-118,193) and the region, which runs from `left flank` over the
    # bubble into `middle`, is corrected hop by hop over its weak anchors (Correction.cpp:609-651, extractSemiWeakPaths :3-157).
    # r11: the 11th error sits ON the SNP (read 320 = hapA 700) with a base that is neither allele: one substitution away from the k-mers of BOTH
    #      bubble branches -> two overlapping variants on different unitigs, keep_non_overlap drops both (Alignment.cpp:1103-1194).
    # r12: the same construction 5 bp further right (no error on the SNP): every weak anchor survives.
    t = hap_a[380:1080]
    def third(c):  # a base that is neither c nor the other allele of the SNP
        return [b for b in "ACGT" if b not in (c, "A", "C")][0]
    R.append(("r11_weak_anchors_snp_conflict", t, edit(t, [(70 + 25 * j, "sub", third(t[70 + 25 * j]) if 70 + 25 * j == 320 else other(t[70 + 25 * j])) for j in range(22)])))
    R.append(("r12_weak_anchors_hops", t, edit(t, [(75 + 25 * j, "sub", other(t[75 + 25 * j])) for j in range(22)])))
    return R


def write_inputs():
    os.makedirs(TOY, exist_ok=True)
    hap_a, hap_b = haplotypes()
    with open(os.path.join(TOY, "hap.fa"), "w") as f:
        f.write(">hapA\n%s\n>hapB\n%s\n" % (hap_a, hap_b))
    with open(os.path.join(TOY, "sr.fq"), "w") as f:
        for n, s in short_reads(hap_a, hap_b):
            f.write("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)))
    reads = long_reads(hap_a, hap_b)
    with open(os.path.join(TOY, "lr.fq"), "w") as f:
        for n, _, s in reads:
            f.write("@%s\n%s\n+\n%s\n" % (n, s, "5" * len(s)))
    with open(os.path.join(TOY, "truth.fa"), "w") as f:
        for n, t, _ in reads:
            f.write(">%s\n%s\n" % (n, t))
    return reads


def build_index(workdir):
    pre = os.path.join(workdir, "toy")
    subprocess.check_call([os.path.join(ROOT, "ratatosk_amd", "bin", "rtk_build_index"), "-s", os.path.join(TOY, "sr.fq"), "-o", pre], stderr=subprocess.DEVNULL)
    return pre


def main():
    import tempfile
    sys.path.insert(0, ROOT)
    from oracle import oracle_py as op
    reads = write_inputs()
    with tempfile.TemporaryDirectory() as d:
        pre = build_index(d)
        og = op.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31)
        print("unitigs:", og.n_unitigs, "k-mers:", og.n_kmers)
        for u in range(og.n_unitigs):
            uu = og.unitig(u)
            print("  unitig %d: %d bp, %d colours, flags %x, fw nb %s, bw nb %s" % (u, len(uu["seq"]), len(uu["local"]), uu["shared"], og.neighbours(u, 0), og.neighbours(u, 1)))
        lr = op.read_fastq(os.path.join(TOY, "lr.fq"))
        want, cnt = og.correct_batch([r[1] for r in lr], [r[2] for r in lr], threads=1)
        print(cnt)
        for (n, truth, raw), (s, q) in zip(reads, want):
            solid, weak = og.seeds(raw)
            print("%-28s raw %4d  out %4d  == truth: %-5s == raw: %-5s  solid %d weak %d  quals %s" % (n, len(raw), len(s), s == truth, s == raw, len(solid), len(weak), "".join(sorted(set(q)))))
        if "--freeze" in sys.argv:
            with open(os.path.join(TOY, "expected.fastq"), "w") as f:
                for (n, _, _), (s, q) in zip(reads, want):
                    f.write("@%s\n%s\n+\n%s\n" % (n, s, q))
            print("froze expected.fastq")


if __name__ == "__main__":
    main()

"""fixSNPs (src/Alignment.cpp:846-965; `-f`, applied to every read of the second pass before phasing(), src/Ratatosk.cpp:828).
Hand-checked cases of the oracle restatement, then the device program (host simulator here, MI355X in the gpu tier) against the
oracle: the stage on its own (rtk_fix_snps) and end to end through `correct -2 -f`."""
import os
import random
import subprocess

import pytest

from conftest import SIM_LIB
from oracle import oracle_py as op
from ratatosk_amd import api
from test_pass2 import ds_pass2, ds_pass2_big  # noqa: F401  (fixtures)

GPU_LIB = None
CODES = "MRSVWYHKDBN"
SETS = {"M": "AC", "R": "AG", "S": "CG", "V": "ACG", "W": "AT", "Y": "CT", "H": "ACT", "K": "GT", "D": "AGT", "B": "CGT", "N": "ACGT"}


def _graphs(pre, lib_path, k):
    fa, rt = pre + ".p2.index.k%d.fasta.gz" % k, pre + ".p2.index.k%d.rtsk" % k
    return op.Graph(fa, rt, k), api.Graph(fa, rt, k, device=0, lib_path=lib_path)


def _quiet_stretch(og, k, need):
    """A stretch of a unitig where no single-base change of any position gives a k-mer of the graph (so that expectations can be
    worked out by hand): returns the string."""
    for u in range(og.n_unitigs):
        s = og.unitig(u)["seq"]
        if len(s) < need:
            continue
        for st in range(0, len(s) - need + 1, 7):
            t = s[st:st + need]
            ok = True
            for p in range(k - 1, need - k + 1):
                for b in "ACGT":
                    if b != t[p]:
                        v = t[:p] + b + t[p + 1:]
                        if any(h >= 0 for h in og.exact(v[p - k + 1:p + k])):
                            ok = False
                            break
                if not ok:
                    break
            if ok:
                return t
    pytest.skip("no variant-free stretch in this graph")


def test_oracle_fix_snps_hand_checked(ds_pass2):
    k = 31
    og = op.Graph(ds_pass2 + ".p2.index.k31.fasta.gz", ds_pass2 + ".p2.index.k31.rtsk", k)
    t = _quiet_stretch(og, k, 4 * k)
    fix = lambda s: og.fix_snps(s).decode()
    put = lambda s, p, c: s[:p] + c + s[p + 1:]
    code_with = lambda b: next(c for c in CODES if b in SETS[c] and len(SETS[c]) == 2)
    assert fix(t) == t                                                            # nothing to do
    p = 2 * k
    assert fix(put(t, p, "N")) == t                                               # one base of four has graph support
    assert fix(put(t, p, code_with(t[p]))) == t                                   # one base of two
    other = next(c for c in CODES if t[p] not in SETS[c])
    assert fix(put(t, p, other)) == put(t, p, other)                              # none of its bases has support: stays
    assert fix(put(t, p, "X")) == put(t, p, "X")                                  # not an IUPAC code: no base at all (:909 never valid)
    assert fix(put(t[:k - 1], 5, "N")) == put(t[:k - 1], 5, "N")                  # read shorter than k (:880)
    assert fix(put(t, 0, "N")) == t and fix(put(t, len(t) - 1, "N")) == t         # first / last character: one-sided windows
    # six two-base codes inside one window: 2^6 = 64 spellings, not "< 64" (:908): every one of them stays
    s6 = t
    for d in range(6):
        s6 = put(s6, p + 3 * d, code_with(t[p + 3 * d]))
    assert fix(s6) == s6
    # five of them: 32 spellings, but the enumeration only runs j < 4 * 5 (:922) with digit d of j = (j >> 2d) & 3 -- the third and
    # later codes only ever get their first base (A for M/R/W, C for S/Y, G for K). The first code resolves iff the others' true
    # bases are reachable that way; work the expectation out with the same rule
    def expect(s):
        s = list(s)
        for i in range(len(s)):
            if s[i] in "ACGT":
                continue
            lo, hi = max(0, i - k + 1), min(i + k, len(s))
            amb = [x for x in range(lo, min(hi + 1, len(s))) if s[x] not in "ACGT"]   # hi inclusive (:893)
            n = 1
            for x in amb:
                n *= len(SETS.get(s[x], ""))
                if n >= 64:
                    break
            if n >= 64:
                continue
            cand = set()
            for j in range(4 * len(amb)):
                if len(cand) > 1:
                    break
                w, valid = list(s), True
                for d, x in enumerate(amb):
                    b = "ACGT"[(j >> (2 * d)) & 3]
                    if b in SETS.get(s[x], ""):
                        w[x] = b
                    else:
                        valid = False
                        break
                if valid and w[i] not in cand and any(h >= 0 for h in og.exact("".join(w[lo:hi]))):
                    cand.add(w[i])
            if len(cand) == 1:
                s[i] = cand.pop()
        return "".join(s)
    s5 = t
    for d in range(5):
        s5 = put(s5, p + 3 * d, code_with(t[p + 3 * d]))
    assert fix(s5) == expect(s5)
    # two N five characters apart in the middle of a read: the k-mers of the first one's window that end before the second one
    # are found whatever the second one is spelled -> both resolve, one after the other
    for q in range(k, 3 * k - 6):
        s2 = put(put(t, q, "N"), q + 5, "N")
        assert fix(s2) == t == expect(s2), q
    # ... at the start of a read every k-mer of the window holds both. While the first N is handled the second one is only ever
    # spelled A or C (j < 8, digit 1 = j >> 2); when the second one is handled it is still the second ambiguity of its window
    seen = set()
    for off in range(0, 2 * k):
        r = t[off:]
        s2 = put(put(r, 2, "N"), 7, "N")
        want = r if r[7] in "AC" else s2
        assert fix(s2) == want == expect(s2), off
        seen.add(r[7] in "AC")
    assert seen == {True, False}
    # an ambiguity right behind the window (position i + k) counts in the number of spellings and takes a digit
    s7 = put(put(t, p, "N"), p + k, "N")
    assert fix(s7) == expect(s7)


def _inject(seqs, k, seed):
    rng = random.Random(seed)
    out = []
    for s in seqs:
        s = list(s)
        if len(s) > 4:
            for _ in range(len(s) // 60):
                s[rng.randrange(len(s))] = rng.choice(CODES)
            for _ in range(3):  # clusters: several codes within one window, foreign characters
                p = rng.randrange(max(1, len(s) - 40))
                for _q in range(rng.randrange(2, 8)):
                    s[min(len(s) - 1, p + rng.randrange(0, 2 * k))] = rng.choice(CODES + "X")
        out.append("".join(s))
    return out


def _check_stage(pre, lib_path, k, n):
    og, pg = _graphs(pre, lib_path, k)
    p1 = op.read_fastq(pre + ".pass1.fq")
    reads = _inject([r[1] for r in p1[:n]], k, 5) + ["", "N", "ACGT" * 5, "N" * 200, "n" * 40 + p1[0][1][:100].lower()]
    resolved = 0
    for s in reads:
        w = og.fix_snps(s.upper())  # the worker upper-cases first (src/Ratatosk.cpp:814)
        assert pg.fix_snps(s) == w
        resolved += sum(1 for a, b in zip(s.upper().encode(), w) if a != b)
    assert resolved > 50


def _check_end_to_end(pre, lib_path, k, n, threads=8):
    og, pg = _graphs(pre, lib_path, k)
    p1, raw = op.read_fastq(pre + ".pass1.fq")[:n], op.read_fastq(pre + ".lr.fq")[:n]
    seqs, quals, raws = _inject([r[1] for r in p1], k, 9), [r[2] for r in p1], [r[1] for r in raw]
    want = og.correct_batch2(seqs, quals, raws, og.opts(force_unres_snp_corr=1), threads=threads)
    got = pg.correct_batch(seqs, quals, pg.opts(long_read_correct=1, force_unres_snp_corr=1), raw=raws)
    assert got == want
    plain = og.correct_batch2(seqs, quals, raws, og.opts(), threads=threads)
    assert sum(1 for a, b in zip(want, plain) if a != b) > 0  # -f made a difference
    return seqs, quals, raws, want


def test_sim_fix_snps_stage(ds_pass2):
    _check_stage(ds_pass2, SIM_LIB, 31, 30)
    _check_stage(ds_pass2, SIM_LIB, 63, 30)


def test_sim_pass2_force_snp(ds_pass2):
    _check_end_to_end(ds_pass2, SIM_LIB, 31, 40)
    _check_end_to_end(ds_pass2, SIM_LIB, 63, 40)


def _check_cli(exe, pre, tmp_path, env, k):
    og = op.Graph(pre + ".p2.index.k%d.fasta.gz" % k, pre + ".p2.index.k%d.rtsk" % k, k)
    p1, raw = op.read_fastq(pre + ".pass1.fq")[:30], op.read_fastq(pre + ".lr.fq")[:30]
    seqs = _inject([r[1] for r in p1], k, 9)
    fin, fraw = str(tmp_path / "in.fq"), str(tmp_path / "raw.fq")
    with open(fin, "w") as f:
        for r, s in zip(p1, seqs):
            f.write("@%s\n%s\n+\n%s\n" % (r[0], s, r[2]))
    with open(fraw, "w") as f:
        for r in raw:
            f.write("@%s\n%s\n+\n%s\n" % (r[0], r[1], r[2]))
    want = og.correct_batch2(seqs, [r[2] for r in p1], [r[1] for r in raw], og.opts(force_unres_snp_corr=1), threads=8)
    out = str(tmp_path / "out")
    subprocess.check_call([exe, "correct", "-2", "-f", "-K", str(k), "-c", "4", "-g", pre + ".p2.index.k%d.fasta.gz" % k, "-d", pre + ".p2.index.k%d.rtsk" % k,
                           "-l", fin, "-L", fraw, "-o", out], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    got = op.read_fastq(out + ".fastq")
    assert [(g[1], g[2]) for g in got] == want
    assert [g[0] for g in got] == [r[0] for r in p1]


def test_sim_cli_force_snp(ds_pass2, tmp_path):
    sim = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "Ratatosk_sim")
    _check_cli(sim, ds_pass2, tmp_path, dict(os.environ, RTK_SIM_DEVICES="1"), 63)


@pytest.mark.gpu
def test_gpu_fix_snps_stage(ds_pass2_big):
    _check_stage(ds_pass2_big, GPU_LIB, 31, 150)
    _check_stage(ds_pass2_big, GPU_LIB, 63, 150)


@pytest.mark.gpu
def test_gpu_pass2_force_snp(ds_pass2_big):
    _check_end_to_end(ds_pass2_big, GPU_LIB, 31, 400, threads=32)
    _check_end_to_end(ds_pass2_big, GPU_LIB, 63, 400, threads=32)


@pytest.mark.gpu
def test_gpu_cli_force_snp(ds_pass2, tmp_path):
    from conftest import BIN
    _check_cli(os.path.join(BIN, "Ratatosk"), ds_pass2, tmp_path, dict(os.environ), 63)

"""Anchor stage (mask, 1-edit k-mer probes, overlap filter, keep_non_overlap, adjacency check) of the device programs,
executed by the host simulator, against the oracle's getSeeds restatement. Exact equality of both anchor lists."""
from conftest import SIM_LIB
from oracle import oracle_py as op
from ratatosk_amd import api


def _check(prefix, n, lib_path):
    fa, rt = prefix + ".index.k31.fasta.gz", prefix + ".index.k31.rtsk"
    og, pg = op.Graph(fa, rt, 31), api.Graph(fa, rt, 31, device=0, lib_path=lib_path)
    reads = op.read_fastq(prefix + ".lr.fq")
    tot_w = 0
    for name, s, q in reads[:n]:
        a, b = pg.seeds(s), og.seeds(s)
        assert a == b, name
        tot_w += len(a[1])
    # edge cases: read of length <= k has no seeds at all (src/Graph.cpp:49); all-N read; read with N inside
    s0 = reads[0][1]
    for s in ["ACGT" * 7 + "ACG", "N" * 100, s0[:300] + "N" + s0[301:900]]:
        assert pg.seeds(s) == og.seeds(s)
    return tot_w


def test_sim_seeds_branching(ds_small):
    assert _check(ds_small, 12, SIM_LIB) > 0  # weak anchors must be exercised


def test_sim_seeds_clean(ds_clean):
    _check(ds_clean, 6, SIM_LIB)


def test_sim_seeds_variant_enumeration_agrees(ds_small, ds_tandem, monkeypatch):
    """The 1-edit search runs on half-k-mer seeds + verification; spelling and probing every variant (RTK_INEXACT_ENUM=1, the first
    implementation) must give the same anchors - also where h-mers repeat (tandem repeats: more than four hits per window)."""
    assert _check(ds_tandem, 10, SIM_LIB) > 0
    monkeypatch.setenv("RTK_INEXACT_ENUM", "1")
    assert _check(ds_small, 6, SIM_LIB) > 0
    assert _check(ds_tandem, 10, SIM_LIB) > 0


def test_sim_seeds_mask_in_segments(ds_small, ds_clean, monkeypatch):
    """k_mask cuts a read into segments of RTK_MASK_SEG windows, one wave each (8192 in production: the launch lasted as long as its longest read).
    With segments of 64 and 192 windows every read of these sets is masked in dozens of pieces: gaps that reach back over a segment border, the read's
    first hit and the last hit before a segment read off the presence bits, head and tail rules applied once. Same anchors as the oracle."""
    for seg in ("64", "192"):
        monkeypatch.setenv("RTK_MASK_SEG", seg)
        assert _check(ds_small, 12, SIM_LIB) > 0
        _check(ds_clean, 6, SIM_LIB)


def _gap_reads(prefix, reps, seed=5):
    """reads built for the mask's rules: stretches of the reference with stretches of random characters between them whose lengths run through the three
    cases of a gap (shorter than half the insert size / between half and the whole of it / longer), + reads without hits at the head, at all, at the tail"""
    import random
    rnd = random.Random(seed)
    junk = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))
    ref = "".join(l.strip() for l in open(prefix + ".ref.fa") if not l.startswith(">"))
    reads = []
    for rep in range(3):
        parts, p = [], rnd.randrange(0, 2000)
        for r2 in range(reps):
            for gap in (60, 130, 240, 250, 260, 300, 380, 460, 499, 500, 501, 640, 900):
                ln = rnd.randrange(40, 330); parts.append(ref[p % (len(ref) - 400):p % (len(ref) - 400) + ln]); p += ln
                parts.append(junk(gap + rnd.randrange(0, 3))); p += gap
        reads.append("".join(parts))
    reads.append(junk(700 * reps) + ref[3000:3400] + junk(300) + ref[3700:3900])   # first hit in a later segment
    reads.append(junk(1500 * reps))                                                 # no hit at all
    reads.append(ref[5000:5300] + junk(1200 * reps))                                # tail without hits
    reads.append(ref[6000:6100] + junk(255) + ref[6355:6400] + junk(251) + ref[6651:6700])
    return reads


def _check_gap_reads(prefixes, reps, segs, lib_path, monkeypatch):
    for prefix in prefixes:
        fa, rt = prefix + ".index.k31.fasta.gz", prefix + ".index.k31.rtsk"
        og = op.Graph(fa, rt, 31)
        reads = _gap_reads(prefix, reps)
        want = [og.seeds(s) for s in reads]
        for seg in segs:
            monkeypatch.setenv("RTK_MASK_SEG", seg)
            pg = api.Graph(fa, rt, 31, device=0, lib_path=lib_path)
            for s, w in zip(reads, want):
                assert pg.seeds(s) == w, (prefix, seg, len(s))


def test_sim_seeds_mask_gaps_across_segment_borders(ds_clean, ds_small, monkeypatch):
    """Reads built for the mask's rules (src/Graph.cpp:102-191) and cut into segments of 64 windows: gaps of every kind start, end and lie across segment
    borders; a read whose first hit comes after several segments without one (the head rule is applied by a later segment), one without any hit, one that
    ends in a long stretch without hits (tail rule)."""
    _check_gap_reads((ds_clean, ds_small), 1, ("64", "128", "8192"), SIM_LIB, monkeypatch)

"""Hand-checked toy fixture (derivations in tests/golden/toy/NOTE.md): a 2 kb diploid genome with one SNP bubble and one
tandem repeat, error-free short reads, thirteen hand-made long reads that walk the branches of correctSequence the note lists.

Three independent legs: (1) facts derived BY HAND from the construction (unitig lengths, corrected sequence = the haplotype substring
the read was made from, the quality patterns of r0 / r1 / r7 / r10) -- no oracle involved; (2) the oracle must reproduce the frozen
expected.fastq; (3) the device programs (simulator here, HIP on the GPU tier) must reproduce it too."""
import os
import sys

import pytest

from conftest import ROOT, SIM_LIB
from oracle import oracle_py as op

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_toy_golden as toy  # noqa: E402

TOY = os.path.join(ROOT, "tests", "golden", "toy")


@pytest.fixture(scope="module")
def toy_index(tmp_path_factory):
    # the committed inputs must be what the generator writes (they are data; the generator documents them)
    hap_a, hap_b = toy.haplotypes()
    assert open(os.path.join(TOY, "hap.fa")).read() == ">hapA\n%s\n>hapB\n%s\n" % (hap_a, hap_b)
    reads = toy.long_reads(hap_a, hap_b)
    assert [(r[0], r[1]) for r in op.read_fastq(os.path.join(TOY, "lr.fq"))] == [(n, raw) for n, _, raw in reads]
    return toy.build_index(str(tmp_path_factory.mktemp("toy"))), reads


def _expected():
    return op.read_fastq(os.path.join(TOY, "expected.fastq"))


def test_toy_graph_is_the_hand_derived_one(toy_index):
    pre, _ = toy_index
    og = op.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31)
    assert og.n_unitigs == 7 and og.n_kmers == 670 + 31 + 31 + 599 + 6 + 6 + 601  # NOTE.md section 1
    assert sorted(len(og.unitig(u)["seq"]) for u in range(7)) == [36, 36, 61, 61, 629, 631, 700]
    from ratatosk_amd import api
    info = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, upload=False).info()
    assert (info.n_unitigs, info.n_kmers) == (7, 1944)


def test_toy_expected_records_match_the_hand_derivation(toy_index):
    """expected.fastq against what NOTE.md derives without running anything."""
    _, reads = toy_index
    exp = _expected()
    assert [e[0] for e in exp] == [r[0] for r in reads]
    for (name, truth, raw), (_, s, q) in zip(reads, exp):
        assert s == truth, name            # every read comes out as the haplotype substring it was made from
        assert len(q) == len(s), name
    q = {e[0]: e[2] for e in exp}
    assert q["r0_all_solid"] == "I" * 800                                   # Correction.cpp:168
    assert q["r1_no_solid"] == "!" * 500                                    # Correction.cpp:170
    assert q["r2_same_unitig_sub"] == "I" * 600                             # Correction.cpp:814-856
    assert q["r7_region_too_long"] == "I" * 150 + "!" * 1100 + "I" * 150    # Correction.cpp:924-930
    # r10: score_best = 1 - 20/107, no second candidate (NOTE.md "r10 by hand"); getQual of src/Common.hpp:410-418
    best = 1.0 - 20.0 / 107.0
    c_best, c_comp = chr(int(best * 40 + 33)), chr(int(best * (40 - 1) + 33 + 1))
    assert c_best == "A" and c_comp == "A"
    assert q["r10_error_cluster_over_snp"] == "I" * 400 + "A" * 46 + "I" * 354
    for name in ("r3_bubble_hapA_del2", "r4_bubble_hapB_ins1", "r5_head_errors", "r6_tail_errors", "r8_revcomp_tandem_hapA", "r9_tandem_hapB_two_errors"):
        assert set(q[name]) == {"I"}, name


def _hand_anchors(first_err, conflict_at=None):
    """r11 / r12 (NOTE.md "r11 and r12 by hand"): 22 substitutions 25 bp apart from read position first_err. Windows without an error are
    solid: [0, first_err - 31] and [first_err + 526, 669]. The 557 windows between them have no exact k-mer (>= insert_sz 500): read
    [first_err, first_err + 526) is searched for 1-edit k-mers. A window holds exactly one error e_j = first_err + 25 j iff it starts in
    [e_j - 24, e_j - 6], and must lie inside the searched stretch: j = 1 .. 20, 19 windows each. conflict_at: the error whose windows are
    one substitution away from k-mers of two unitigs (the SNP bubble) and are dropped by keep_non_overlap."""
    solid = list(range(0, first_err - 31 + 1)) + list(range(first_err + 526, 670))
    weak = []
    for j in range(1, 21):
        e = first_err + 25 * j
        if e == conflict_at:
            continue
        weak += list(range(e - 24, e - 6 + 1))
    return solid, weak


def _check_hand_anchors(seeds_of, og, reads):
    rc = lambda x: x[::-1].translate(str.maketrans("ACGT", "TGCA"))
    by = {r[0]: r for r in reads}
    for name, first_err, conflict in (("r11_weak_anchors_snp_conflict", 70, 320), ("r12_weak_anchors_hops", 75, None)):
        _, truth, raw = by[name]
        solid, weak = seeds_of(raw)
        h_solid, h_weak = _hand_anchors(first_err, conflict)
        assert [a[0] for a in solid] == h_solid, name
        assert [a[0] for a in weak] == h_weak, name
        assert len(h_weak) == (361 if conflict else 380)
        for pos, u, dist, strand in weak:  # every weak anchor is the k-mer of the haplotype the read was made from: the read window with its one error undone
            useq = og.unitig(u)["seq"]
            km = useq[dist:dist + 31]
            assert (km if strand else rc(km)) == truth[pos:pos + 31], (name, pos)


def test_toy_weak_anchor_lists_are_the_hand_derived_ones(toy_index):
    """getSeeds of r11 / r12 against lists written down from the construction alone: which windows are solid, which stretch is searched for
    1-edit k-mers, which windows have a hit, and which hits keep_non_overlap drops (the error on the SNP: hits on both bubble branches)."""
    pre, reads = toy_index
    og = op.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31)
    _check_hand_anchors(og.seeds, og, reads)
    from ratatosk_amd import api
    pg = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, device=0, lib_path=SIM_LIB)
    _check_hand_anchors(pg.seeds, og, reads)


@pytest.mark.gpu
def test_gpu_toy_weak_anchor_lists_are_the_hand_derived_ones(toy_index):
    pre, reads = toy_index
    og = op.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31)
    from ratatosk_amd import api
    pg = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, device=0)
    _check_hand_anchors(pg.seeds, og, reads)


def test_toy_oracle_reproduces_expected(toy_index):
    pre, reads = toy_index
    og = op.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31)
    want, _ = og.correct_batch([r[2] for r in reads], ["5" * len(r[2]) for r in reads], threads=2)
    assert want == [(e[1], e[2]) for e in _expected()]


def _device(pre, reads, lib_path):
    from ratatosk_amd import api
    pg = api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, device=0, lib_path=lib_path)
    return pg.correct_batch([r[2] for r in reads], ["5" * len(r[2]) for r in reads])


def test_toy_device_program_on_simulator_reproduces_expected(toy_index):
    pre, reads = toy_index
    assert _device(pre, reads, SIM_LIB) == [(e[1], e[2]) for e in _expected()]


@pytest.mark.gpu
def test_gpu_toy_reproduces_expected(toy_index):
    pre, reads = toy_index
    assert _device(pre, reads, None) == [(e[1], e[2]) for e in _expected()]


# ---- the [A3] toy (tests/golden/toy_a3/NOTE.md): a tie between the two branches of a bubble that the order of getSuccessors() decides ----
import gen_toy_a3 as toy3  # noqa: E402


@pytest.fixture(scope="module")
def toy_a3(tmp_path_factory):
    hap_a, hap_b = toy3.haplotypes()
    assert open(os.path.join(toy3.OUT, "hap.fa")).read() == ">hapA\n%s\n>hapB\n%s\n" % (hap_a, hap_b)
    reads = toy3.long_reads(hap_a)
    assert [(r[0], r[1]) for r in op.read_fastq(os.path.join(toy3.OUT, "lr.fq"))] == [(n, raw) for n, raw, _ in reads]
    pre = toy3.build_index(str(tmp_path_factory.mktemp("toy_a3")))
    og = op.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31)
    seqs = [og.unitig(u)["seq"] for u in range(og.n_unitigs)]
    assert sorted(len(s) for s in seqs) == [61, 61, 599, 600]  # NOTE.md "Graph"
    left = [s for s in seqs if len(s) == 600][0]
    assert left == hap_a[:600] or left == toy3.rc(hap_a[:600])
    return pre, reads, left == hap_a[:600]


def _a3_expected(reads, left_flank_in_hap_orientation, reading):
    """NOTE.md "Which branch is listed last": the corrected base by hand"""
    base = "C" if (left_flank_in_hap_orientation or reading == "walk") else "A"
    return [(raw[:p] + base + raw[p + 1:], p) for _, raw, p in reads]


def _check_a3(correct, toy_a3, monkeypatch):
    pre, reads, left_fw = toy_a3
    outs = {}
    for reading in ("walk", "strand"):
        monkeypatch.setenv("RTK_A3_ORDER", reading)
        got = correct(pre, [r[1] for r in reads], ["5" * len(r[1]) for r in reads])
        for (s, q), (want, p) in zip(got, _a3_expected(reads, left_fw, reading)):
            assert s == want, (reading, "corrected base at the SNP: %s, by hand: %s" % (s[p], want[p]))
            assert q[:p] == "I" * p and q[p + 1:] == "I" * (len(s) - p - 1) and q[p] != "I"
        outs[reading] = got
    monkeypatch.delenv("RTK_A3_ORDER", raising=False)
    assert not left_fw and outs["walk"] != outs["strand"]  # the fixture separates the two readings (the tool writes the left flank reverse-complemented)
    assert correct(pre, [r[1] for r in reads], ["5" * len(r[1]) for r in reads]) == outs["walk"]  # the default is the reading argued for in the note


def test_toy_a3_oracle_and_simulator_take_the_hand_derived_branch(toy_a3, monkeypatch):
    from ratatosk_amd import api
    _check_a3(lambda pre, s, q: op.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31).correct_batch(s, q)[0], toy_a3, monkeypatch)
    _check_a3(lambda pre, s, q: api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, device=0, lib_path=SIM_LIB).correct_batch(s, q), toy_a3, monkeypatch)


@pytest.mark.gpu
def test_gpu_toy_a3_takes_the_hand_derived_branch(toy_a3, monkeypatch):
    from ratatosk_amd import api
    _check_a3(lambda pre, s, q: api.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31, device=0).correct_batch(s, q), toy_a3, monkeypatch)

"""Second correction pass (`correct -2`): phasing() pre-filter (src/Graph.cpp:869-1097), exact-only seeds (src/Graph.cpp:100),
exploreSubGraphLong (src/GraphTraversal.cpp:589-720), quality carry-over and hasMinQual skips (src/Correction.cpp:779,808,941).
Device programs (host simulator here, MI355X in the gpu tier) against the oracle, byte for byte."""
import os
import subprocess

import pytest

from conftest import BIN, SIM_LIB, make_dataset
from oracle import oracle_py as op
from ratatosk_amd import api

GPU_LIB = None  # default library of ratatosk_amd.api


def _second_pass_set(tmpdir, name, sim_args):
    """Pass 1 by the oracle (so the fixture needs no GPU), then the second-pass index: graph, coverage and colours from the
    pass-1 reads, colour = read index (src/Ratatosk.cpp:1079-1101)."""
    pre = make_dataset(tmpdir, name, sim_args)
    og = op.Graph(pre + ".index.k31.fasta.gz", pre + ".index.k31.rtsk", 31)
    raw = op.read_fastq(pre + ".lr.fq")
    out, _ = og.correct_batch([r[1] for r in raw], [r[2] for r in raw], threads=8)
    p1 = pre + ".pass1.fq"
    with open(p1, "w") as f:
        for r, (s, q) in zip(raw, out):
            f.write("@%s\n%s\n+\n%s\n" % (r[0], s, q))
    subprocess.check_call([os.path.join(BIN, "rtk_build_index"), "-s", p1, "--colour-reads", p1, "-o", pre + ".p2"], stderr=subprocess.DEVNULL)
    return pre


@pytest.fixture(scope="module")
def ds_pass2(tmp_path_factory):
    return _second_pass_set(tmp_path_factory.mktemp("ds_pass2"), "p2", ["--seed", 31, "--ref-len", 40000, "--het", 0.004, "--repeat-frac", 0.05, "--sr-cov", 40, "--sr-err", 0.005,
                                                                        "--lr-n", 60, "--lr-len", 3000, "--lr-profile", "ont", "--lr-err", 0.08])


@pytest.fixture(scope="module")
def ds_pass2_big(tmp_path_factory):
    return _second_pass_set(tmp_path_factory.mktemp("ds_pass2_big"), "p2b", ["--seed", 33, "--ref-len", 300000, "--het", 0.003, "--repeat-frac", 0.05, "--sr-cov", 30, "--sr-err", 0.005,
                                                                             "--lr-n", 400, "--lr-len", 6000, "--lr-profile", "ont", "--lr-err", 0.08])


def _load(pre, lib_path):
    fa, rt = pre + ".p2.index.k31.fasta.gz", pre + ".p2.index.k31.rtsk"
    p1, raw = op.read_fastq(pre + ".pass1.fq"), op.read_fastq(pre + ".lr.fq")
    return op.Graph(fa, rt, 31), api.Graph(fa, rt, 31, device=0, lib_path=lib_path), [r[1] for r in p1], [r[2] for r in p1], [r[1] for r in raw]


def _check_pass2(pre, lib_path, n=None, threads=8):
    og, pg, seqs, quals, raws = _load(pre, lib_path)
    if n:
        seqs, quals, raws = seqs[:n], quals[:n], raws[:n]
    oo = og.opts(long_read_correct=1)
    want = og.correct_batch2(seqs, quals, raws, oo, threads=threads)
    got = pg.correct_batch(seqs, quals, pg.opts(long_read_correct=1), raw=raws)
    for i, (g_, w_) in enumerate(zip(got, want)):
        assert g_[0] == w_[0], "sequence of read %d differs" % i
        assert g_[1] == w_[1], "quality of read %d differs" % i
    # the pre-filter must have had something to do, and so must the pass itself
    no_pre = og.correct_batch2(seqs, quals, raws, og.opts(long_read_correct=1, skip_phasing=1), threads=threads)
    assert sum(1 for a, b in zip(want, no_pre) if a != b) > 0
    assert sum(1 for (s, _), s0 in zip(want, seqs) if s != s0) > 0
    return og, pg, seqs, quals, raws


def test_sim_pass2_matches_oracle(ds_pass2):
    og, pg, seqs, quals, raws = _check_pass2(ds_pass2, SIM_LIB)
    # without the uncorrected reads the pre-filter cannot run: same as the oracle with phasing() left out
    want = og.correct_batch2(seqs[:20], quals[:20], raws[:20], og.opts(long_read_correct=1, skip_phasing=1), threads=4)
    got = pg.correct_batch(seqs[:20], quals[:20], pg.opts(long_read_correct=1))
    assert got == want


def test_sim_pass2_edge_reads(ds_pass2):
    """Reads phasing() leaves alone or empties: shorter than k, all N, raw read unrelated to the corrected one, empty raw read."""
    og, pg, seqs, quals, raws = _load(ds_pass2, SIM_LIB)
    s, q, r = list(seqs[:4]), list(quals[:4]), list(raws[:4])
    s += ["ACGT" * 5, "N" * 100, seqs[5], seqs[6][:50], seqs[8], ""]
    q += ["I" * 20, "!" * 100, quals[5], quals[6][:50], quals[8], ""]
    r += ["ACGT" * 5, "N" * 90, raws[7], raws[6][:40], "", raws[9]]
    want = og.correct_batch2(s, q, r, og.opts(long_read_correct=1), threads=2)
    got = pg.correct_batch(s, q, pg.opts(long_read_correct=1), raw=r)
    assert got == want


def test_sim_pass2_needs_qualities(ds_pass2):
    og, pg, seqs, quals, raws = _load(ds_pass2, SIM_LIB)
    with pytest.raises(Exception):
        pg.correct_batch(seqs[:2], None, pg.opts(long_read_correct=1))


@pytest.mark.gpu
def test_gpu_pass2_matches_oracle(ds_pass2_big):
    _check_pass2(ds_pass2_big, GPU_LIB, threads=32)


@pytest.mark.gpu
def test_gpu_pass2_small(ds_pass2):
    _check_pass2(ds_pass2, GPU_LIB)

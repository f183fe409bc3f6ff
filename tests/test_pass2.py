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
    # the reference's second graph is the short-read graph at k2, coloured by the pass-1 reads (src/Ratatosk.cpp:1193,1227)
    subprocess.check_call([os.path.join(BIN, "rtk_build_index"), "-s", pre + ".sr.fq", "--colour-reads", p1, "-o", pre + ".p2"], stderr=subprocess.DEVNULL)
    # ... and at the reference's default k2 = 63 (src/Common.hpp:101): two-word k-mers
    subprocess.check_call([os.path.join(BIN, "rtk_build_index"), "-s", pre + ".sr.fq", "--colour-reads", p1, "-k", "63", "-o", pre + ".p2"], stderr=subprocess.DEVNULL)
    return pre


@pytest.fixture(scope="module")
def ds_pass2(tmp_path_factory):
    return _second_pass_set(tmp_path_factory.mktemp("ds_pass2"), "p2", ["--seed", 31, "--ref-len", 40000, "--het", 0.004, "--repeat-frac", 0.05, "--sr-cov", 40, "--sr-err", 0.005,
                                                                        "--lr-n", 60, "--lr-len", 3000, "--lr-profile", "ont", "--lr-err", 0.08])


@pytest.fixture(scope="module")
def ds_pass2_big(tmp_path_factory):
    return _second_pass_set(tmp_path_factory.mktemp("ds_pass2_big"), "p2b", ["--seed", 33, "--ref-len", 300000, "--het", 0.003, "--repeat-frac", 0.05, "--sr-cov", 30, "--sr-err", 0.005,
                                                                             "--lr-n", 400, "--lr-len", 6000, "--lr-profile", "ont", "--lr-err", 0.08])


def _load(pre, lib_path, k=31):
    fa, rt = pre + ".p2.index.k%d.fasta.gz" % k, pre + ".p2.index.k%d.rtsk" % k
    p1, raw = op.read_fastq(pre + ".pass1.fq"), op.read_fastq(pre + ".lr.fq")
    return op.Graph(fa, rt, k), api.Graph(fa, rt, k, device=0, lib_path=lib_path), [r[1] for r in p1], [r[2] for r in p1], [r[1] for r in raw]


def _check_pass2(pre, lib_path, n=None, threads=8, k=31):
    og, pg, seqs, quals, raws = _load(pre, lib_path, k)
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



def _skip_check(pre, lib_path, monkeypatch, n=None, k=31, both_kinds=True):
    """phasing(): a read without an unsupported stretch comes back as it is whatever its alignment against the raw read looks like (src/Graph.cpp:975-1069 with an
    empty pos2rm), so the device skips that alignment. Same bytes with the skip and with every read aligned (RTK_PHASE_ALIGN_ALL=1), both kinds of read present."""
    _, pg, seqs, quals, raws = _load(pre, lib_path, k)
    if n:
        seqs, quals, raws = seqs[:n], quals[:n], raws[:n]
    o = pg.opts(long_read_correct=1)
    b = api.Batch(pg, seqs, quals, raw=raws); b.run(o); got = b.fetch(); st = b.stats()
    assert (0 < st["n_phase_skipped"] < len(seqs)) or not both_kinds, st["n_phase_skipped"]
    monkeypatch.setenv("RTK_PHASE_ALIGN_ALL", "1")
    b2 = api.Batch(pg, seqs, quals, raw=raws); b2.run(o); got_all = b2.fetch()
    assert b2.stats()["n_phase_skipped"] == 0
    assert got == got_all
    return st["n_phase_skipped"], len(seqs)


def test_sim_pass2_alignment_skipped_where_it_cannot_matter(ds_pass2, monkeypatch):
    _skip_check(ds_pass2, SIM_LIB, monkeypatch)


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


def _check_lookup_k63(pre, lib_path):
    """Exact lookups with two-word k-mers ([A1]): bit-exact (unitig, dist, strand) for reads, both strands of a unitig, N inside."""
    og, pg, seqs, quals, raws = _load(pre, lib_path, 63)
    u = og.unitig(0)["seq"]
    for s in seqs[:6] + raws[:2] + ["ACGT" * 5, "A" * 63, seqs[0][:200] + "N" + seqs[0][200:400], u, op.revcomp(u) if hasattr(op, "revcomp") else u[::-1].translate(str.maketrans("ACGT", "TGCA"))]:
        assert pg.lookup_exact(s) == og.exact(s)
    hits = pg.lookup_exact(u)
    assert len(hits) == len(u) - 62 and all(h >= 0 and (h & 1) for h in hits) and [((h >> 1) & 0xFFFFFFFF) for h in hits] == list(range(len(hits)))
    assert pg.info().n_kmers == og.n_kmers
    with pytest.raises(api.RtkError):  # the first pass (1-edit search) is limited to one-word k-mers
        pg.correct_batch(seqs[:1], None, pg.opts())


def test_sim_k63_lookup(ds_pass2):
    _check_lookup_k63(ds_pass2, SIM_LIB)


def test_sim_pass2_k63_matches_oracle(ds_pass2):
    _check_pass2(ds_pass2, SIM_LIB, k=63)


def test_sim_pass2_k63_volume(ds_pass2_big):
    """400 reads: reaches the cases the small set does not (a reverted base more than 32 positions away from the window that
    restores its quality -- found on the GPU tier first)."""
    _check_pass2(ds_pass2_big, SIM_LIB, threads=16, k=63)


def test_sim_cli_pass2_k63(ds_pass2, tmp_path):
    sim = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "Ratatosk_sim")
    _check_cli_pass2(sim, ds_pass2, tmp_path, dict(os.environ, RTK_SIM_DEVICES="1"), k=63)


@pytest.mark.gpu
def test_gpu_pass2_matches_oracle(ds_pass2_big):
    _check_pass2(ds_pass2_big, GPU_LIB, threads=32)


@pytest.mark.gpu
def test_gpu_pass2_k63_matches_oracle(ds_pass2_big):
    _check_lookup_k63(ds_pass2_big, GPU_LIB)
    _check_pass2(ds_pass2_big, GPU_LIB, threads=32, k=63)


@pytest.fixture(scope="module")
def ds_pass2_long(tmp_path_factory):
    """A few very long reads (up to ~90 kb: more than 16 row blocks of 4096 query rows in the whole-read alignment of phasing())."""
    return _second_pass_set(tmp_path_factory.mktemp("ds_pass2_long"), "p2l", ["--seed", 35, "--ref-len", 400000, "--het", 0.002, "--sr-cov", 30, "--sr-err", 0.005,
                                                                              "--lr-n", 12, "--lr-len", 45000, "--lr-profile", "ont", "--lr-err", 0.07])


@pytest.mark.gpu
def test_gpu_pass2_multiwave_kernel(ds_pass2_big, ds_pass2_long, monkeypatch):
    """The multi-wave phasing kernel (rtk_phase_long.hip: the row blocks of one alignment pass on the waves of a workgroup) against
    the oracle: forced onto every read longer than 2 kb of the 6 kb set (two row blocks per pass), then on reads of tens of kb with its
    default threshold (several blocks per wave)."""
    monkeypatch.setenv("RTK_PHASE_LONG", "2000")
    _check_pass2(ds_pass2_big, GPU_LIB, threads=32, k=63)
    monkeypatch.delenv("RTK_PHASE_LONG")
    og, pg, seqs, quals, raws = _load(ds_pass2_long, GPU_LIB, 31)
    assert max(len(r) for r in raws) > 70000
    want = og.correct_batch2(seqs, quals, raws, og.opts(long_read_correct=1), threads=12)
    got = pg.correct_batch(seqs, quals, pg.opts(long_read_correct=1), raw=raws)
    assert got == want
    monkeypatch.setenv("RTK_PHASE_LONG", "0")  # the same reads on one wave each
    assert pg.correct_batch(seqs, quals, pg.opts(long_read_correct=1), raw=raws) == want


@pytest.mark.gpu
def test_gpu_pass2_alignment_skipped_or_pruned(ds_pass2_big, ds_pass2_long, monkeypatch):
    """The same on the device, whose level-by-level Hirschberg driver (rtk_myers_lvl.h) and multi-wave kernel have their own pruning code: 6 kb reads on one wave each,
    the same reads on the multi-wave kernel, reads of tens of kb."""
    _skip_check(ds_pass2_big, GPU_LIB, monkeypatch, k=63)
    monkeypatch.delenv("RTK_PHASE_ALIGN_ALL")
    monkeypatch.setenv("RTK_PHASE_LONG", "2000")
    _skip_check(ds_pass2_big, GPU_LIB, monkeypatch)
    monkeypatch.delenv("RTK_PHASE_ALIGN_ALL"); monkeypatch.delenv("RTK_PHASE_LONG")
    _skip_check(ds_pass2_long, GPU_LIB, monkeypatch, both_kinds=False)


@pytest.mark.gpu
def test_gpu_pass2_small(ds_pass2):
    _check_pass2(ds_pass2, GPU_LIB)


def _trim_records(name, seq, qual, k, trim):
    """writeCorrectedOutput (src/Ratatosk.cpp:508-563), restated for the test."""
    if trim == 0:
        return [(name, seq, qual)]
    out, start, run = [], -1, -1
    for pos, c in enumerate(qual):
        if ord(c) >= trim + 33:
            if start == -1:
                start, run = pos, 0
            run += 1
        else:
            if run >= k:
                out.append(("%s/%d" % (name, len(out) + 1), seq[start:start + run], qual[start:start + run]))
            start, run = -1, -1
    if run >= k:
        out.append(("%s/%d" % (name, len(out) + 1), seq[start:start + run], qual[start:start + run]))
    return out


def _cli_pass2(exe, pre, tmp_path, extra, env=None, k=31):
    out = str(tmp_path / "out")
    r = subprocess.run([exe, "correct", "-2"] + (["-K", str(k)] if k != 63 else []) + ["-c", "2", "-B", "20000", "-g", pre + ".p2.index.k%d.fasta.gz" % k, "-d", pre + ".p2.index.k%d.rtsk" % k,
                        "-l", pre + ".pass1.fq", "-L", pre + ".lr.fq", "-o", out] + extra, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    return out


def _expected_pass2(pre, k=31):
    og, _, seqs, quals, raws = _load(pre, SIM_LIB, k)
    names = [r[0] for r in op.read_fastq(pre + ".pass1.fq")]
    return names, og.correct_batch2(seqs, quals, raws, og.opts(long_read_correct=1), threads=8)


def _check_cli_pass2(exe, pre, tmp_path, env=None, k=31):
    names, want = _expected_pass2(pre, k)
    out = _cli_pass2(exe, pre, tmp_path, [], env, k)
    got = op.read_fastq(out + ".fastq")
    assert [(g[1], g[2]) for g in got] == want and [g[0] for g in got] == names
    # -t 20 -G: trimmed / split records, gzip members per ticket block
    out = _cli_pass2(exe, pre, tmp_path, ["-t", "20", "-G"], env, k)
    got = op.read_fastq(out + ".fastq.gz")
    exp = [rec for n_, (s, q) in zip(names, want) for rec in _trim_records(n_, s, q, k, 20)]
    assert got == exp and 0 < len(exp) and any("/2" in e[0] for e in exp)


def test_sim_cli_pass2(ds_pass2, tmp_path):
    sim = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "Ratatosk_sim")
    env = dict(os.environ, RTK_SIM_DEVICES="2")
    _check_cli_pass2(sim, ds_pass2, tmp_path, env)
    # reads out of step with the uncorrected file: the reference aborts (src/Ratatosk.cpp:787-797)
    raw = op.read_fastq(ds_pass2 + ".lr.fq")
    swapped = str(tmp_path / "swapped.fq")
    with open(swapped, "w") as f:
        for r in [raw[1], raw[0]] + raw[2:]:
            f.write("@%s\n%s\n+\n%s\n" % r)
    r = subprocess.run([sim, "correct", "-2", "-K", "31", "-g", ds_pass2 + ".p2.index.k31.fasta.gz", "-d", ds_pass2 + ".p2.index.k31.rtsk", "-l", ds_pass2 + ".pass1.fq",
                        "-L", swapped, "-o", str(tmp_path / "bad")], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "not in the same order" in r.stderr and not os.path.exists(str(tmp_path / "bad.fastq"))


@pytest.mark.gpu
def test_gpu_cli_pass1_then_pass2(ds_pass2, tmp_path):
    """`correct -1` then `correct -2` through the executable: both files equal the oracle's."""
    exe = os.path.join(BIN, "Ratatosk")
    out = str(tmp_path / "p1")
    r = subprocess.run([exe, "correct", "-1", "-c", "2", "-g", ds_pass2 + ".index.k31.fasta.gz", "-d", ds_pass2 + ".index.k31.rtsk", "-l", ds_pass2 + ".lr.fq", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out + ".2.fastq").read() == open(ds_pass2 + ".pass1.fq").read()
    _check_cli_pass2(exe, ds_pass2, tmp_path)
    _check_cli_pass2(exe, ds_pass2, tmp_path, k=63)

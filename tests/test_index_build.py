"""Index build. (1) Against an INDEPENDENT oracle (oracle/oracle_index.py: dictionary k-mer count, unitigs by the join relation followed from the
path starts, colours by pair id, coverage): the unitigs (up to strand / rotation), colour sets and coverages of the files the tool writes -- plain,
`--fast` and, on the GPU tier, `--gpu` -- are the oracle's, on seeded sets with heterozygous SNPs, repeats and tandem repeats, k = 31 and k = 21, and
for a second-pass index. (2) Fast paths: `rtk_build_index --fast` (thread-parallel table fill, unitig construction, adjacency, cycle search)
and `--gpu` (--fast with the k-mers counted on the device: csrc/hip/rtk_index.hip, rtk_index_count_kmers) must write the SAME two files as the
plain single-path tool, byte for byte -- the plain tool is their oracle. Seeded sets with heterozygous SNPs, two-copy repeats and tandem
repeats, k = 31 and k = 21, and two hand-made genomes whose chains of k-mers meet themselves (a closed loop, a hairpin): those take the
plain construction inside the fast path."""
import os
import subprocess

import pytest

from conftest import BIN

SETS = [
    ("het_repeats", ["--seed", "11", "--ref-len", "30000", "--het", "0.004", "--repeat-frac", "0.1", "--sr-cov", "40", "--sr-err", "0.01"]),
    ("tandem", ["--seed", "21", "--ref-len", "60000", "--het", "0.003", "--tandem", "30", "--sr-cov", "40", "--sr-err", "0.005"]),
    ("diploid_400k", ["--seed", "7", "--ref-len", "400000", "--het", "0.002", "--repeat-frac", "0.05", "--sr-cov", "30", "--sr-err", "0.005"]),
    # k-mers of tandem repeats with dozens of one-substitution neighbours at k = 21 (the fast SNP search kept 16 per window and lost the rest:
    # found by the second implementation of the annotators, tests/test_annotators.py)
    ("many_neighbours", ["--seed", "624414", "--ref-len", "90000", "--het", "0.002", "--repeat-frac", "0.05", "--tandem", "5", "--sr-cov", "60", "--sr-err", "0.01"]),
]


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def _build(sr, out, k, extra):
    r = subprocess.run([os.path.join(BIN, "rtk_build_index"), "-s", sr, "-o", out, "-k", str(k), "--snps"] + extra, capture_output=True, text=True, env=dict(os.environ, RTK_INDEX_TRACE="1"))
    assert r.returncode == 0, r.stderr
    if "--gpu" in extra:  # the chains were walked and written on the device (rtk_index_unitigs), not on the host threads behind a failed call
        assert "rtk_index_unitigs:" in r.stderr and "unitigs on the host threads" not in r.stderr, r.stderr
        assert "rtk_index_colour:" in r.stderr and "colours on the host threads" not in r.stderr, r.stderr  # ... and the reads mapped there (rtk_index_colour_*)
    return open(out + ".index.k%d.fasta.gz" % k, "rb").read(), open(out + ".index.k%d.rtsk" % k, "rb").read(), r.stderr


def _same(tmp, name, sr, mode, ks=(31, 21)):
    for k in ks:
        a = _build(sr, os.path.join(tmp, name + "_plain"), k, [])
        b = _build(sr, os.path.join(tmp, name + "_" + mode.strip("-")), k, [mode])
        assert a[0] == b[0], (name, k, "unitig FASTA differs")
        assert a[1] == b[1], (name, k, ".rtsk differs")
    return b[2]


def _simulated(tmp, name, args):
    pre = os.path.join(tmp, name)
    subprocess.check_call([os.path.join(BIN, "rtk_simulate"), "--prefix", pre] + args + ["--lr-n", "2", "--lr-len", "1000"], stderr=subprocess.DEVNULL)
    return pre + ".sr.fq"


def _self_meeting_genomes(tmp):
    """Reads tiling (a) a circular 600 bp sequence -- every k-mer has one successor and one predecessor: a closed loop without an end --
    and (b) W + reverse complement of W: the chain of k-mers runs into its own reverse complement (hairpin)."""
    import random
    rnd = random.Random(5)
    circ = "".join(rnd.choice("ACGT") for _ in range(600))
    w = "".join(rnd.choice("ACGT") for _ in range(300))
    hair = w + _rc(w)
    sr = os.path.join(tmp, "self.sr.fq")
    with open(sr, "w") as f:
        n = 0
        for rep in range(3):
            for start in range(0, 600, 7):
                s = (circ + circ)[start:start + 100]
                f.write("@c%d\n%s\n+\n%s\n" % (n, s, "I" * 100)); n += 1
            for start in range(0, len(hair) - 100 + 1, 5):
                s = hair[start:start + 100]
                f.write("@h%d\n%s\n+\n%s\n" % (n, s, "I" * 100)); n += 1
    return sr


ORACLE_SETS = [  # small enough for a Python dictionary of every read k-mer
    ("o_het_repeats", ["--seed", "11", "--ref-len", "30000", "--het", "0.004", "--repeat-frac", "0.1", "--sr-cov", "30", "--sr-err", "0.01"]),
    ("o_tandem", ["--seed", "21", "--ref-len", "40000", "--het", "0.003", "--tandem", "20", "--sr-cov", "25", "--sr-err", "0.005"]),
    ("o_diploid", ["--seed", "7", "--ref-len", "90000", "--het", "0.002", "--repeat-frac", "0.05", "--sr-cov", "20", "--sr-err", "0.005"]),
]


def _index_vs_oracle(sr, out, k, mode, colour=None):
    """the unitigs / colours / coverages of the files `rtk_build_index <mode>` writes against oracle/oracle_index.py"""
    from oracle import oracle_index as oi
    from oracle import oracle_py as op
    cmd = [os.path.join(BIN, "rtk_build_index"), "-s", sr, "-o", out, "-k", str(k)] + (["--colour-reads", colour] if colour else []) + mode
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want, solid = oi.build([sr], k, 2, [colour] if colour else None)
    g = op.Graph(out + ".index.k%d.fasta.gz" % k, out + ".index.k%d.rtsk" % k, k)
    assert g.n_kmers == len(solid), (g.n_kmers, len(solid))
    got = {}
    for u in range(g.n_unitigs):
        d = g.unitig(u)
        ids = sorted(set(d["local"]) | (set(g.global_set(d["global_id"])) if d["global_id"] >= 0 else set()))
        got[oi.canonical(d["seq"])] = (ids, (d["kmcov"] >> 31) & 0x7FFFFFFF, d["seq"])
    assert len(got) == g.n_unitigs == len(want), (g.n_unitigs, len(want))
    missing = [key for key in want if key not in got]
    if missing:  # isolated cycles: the tool opens them elsewhere; compare by their rotation-independent form
        rot = {oi.unitig_key(v[2], k, True): v for key, v in got.items() if key not in want}
        for key in missing:
            assert key in rot, ("unitig of the oracle not in the index", key[:80])
            got[key] = rot[key]
    bad = [(key[:50], want[key][1], got[key][1]) for key in want if want[key][1] != got[key][1]]
    assert not bad, ("coverage differs", bad[:5])
    bad = [key[:50] for key in want if want[key][0] != got[key][0]]
    assert not bad, ("colour sets differ", bad[:5])
    return len(want)


def test_index_build_against_the_independent_oracle(tmp_path):
    tmp = str(tmp_path)
    for name, args in ORACLE_SETS:
        sr = _simulated(tmp, name, args)
        for k in (31, 21):
            n = _index_vs_oracle(sr, os.path.join(tmp, name + "_plain"), k, [])
            assert n > 10
            assert _index_vs_oracle(sr, os.path.join(tmp, name + "_fast"), k, ["--fast"]) == n
    # second-pass index: the graph of the short reads coloured by the long reads, every read its own id (src/Ratatosk.cpp:1218)
    sr = _simulated(tmp, "o_p2", ORACLE_SETS[0][1])
    lr = os.path.join(tmp, "o_p2.lr.fq")
    for mode in ([], ["--fast"]):
        _index_vs_oracle(sr, os.path.join(tmp, "o_p2_" + ("fast" if mode else "plain")), 31, mode, colour=lr)
    # chains of k-mers that meet themselves: a closed loop (compared up to rotation) and a hairpin
    _index_vs_oracle(_self_meeting_genomes(tmp), os.path.join(tmp, "o_self"), 31, [])
    _index_vs_oracle(_self_meeting_genomes(tmp), os.path.join(tmp, "o_self_f"), 31, ["--fast"])


@pytest.mark.gpu
def test_gpu_index_build_against_the_independent_oracle(tmp_path):
    """`--gpu` (k-mers counted on the device) held to oracle/oracle_index.py, not to the repo's own plain tool"""
    tmp = str(tmp_path)
    for name, args in ORACLE_SETS:
        sr = _simulated(tmp, name, args)
        for k in (31, 21):
            assert _index_vs_oracle(sr, os.path.join(tmp, name + "_gpu"), k, ["--gpu"]) > 10
    _index_vs_oracle(_self_meeting_genomes(tmp), os.path.join(tmp, "o_self_g"), 31, ["--gpu"])


def _sample_spec(tmp, name):
    """a reference written by rtk_simulate (two haplotypes) + the `sample:` source over it (common/sample_source.hpp)"""
    pre = os.path.join(tmp, name)
    subprocess.check_call([os.path.join(BIN, "rtk_simulate"), "--prefix", pre, "--seed", "4", "--ref-len", "120000", "--het", "0.002", "--repeat-frac", "0.05", "--sr-cov", "0", "--lr-n", "2", "--lr-len", "1000"], stderr=subprocess.DEVNULL)
    return "sample:%s.ref.fa?cov=24&len=150&insert=400&err=0.005&seed=9" % pre


def test_index_from_reads_sampled_on_the_fly(tmp_path):
    """`-s sample:REF.fa?cov=..` (short reads generated inside the tool, pair by pair from (seed, pair number): the input of the whole-genome-scale set
    never exists as a file) gives the index of the FASTQ file it stands for (`--dump-input`), through the plain and the thread-parallel paths."""
    tmp = str(tmp_path)
    spec = _sample_spec(tmp, "smp")
    dump = os.path.join(tmp, "smp.dump.fq")
    subprocess.check_call([os.path.join(BIN, "rtk_build_index"), "-s", spec, "--dump-input", dump])
    lines = open(dump).read().split("\n")
    assert len(lines) // 4 == 2 * int(24 * 120000 / 300) and lines[0] == lines[4] == "@s0" and lines[8] == "@s1" and len(lines[1]) == 150 and lines[1] != lines[5]
    want = _build(dump, os.path.join(tmp, "smp_file"), 31, [])
    for mode in ([], ["--fast"]):
        got = _build(spec, os.path.join(tmp, "smp_src"), 31, mode)
        assert got[0] == want[0] and got[1] == want[1], mode
    assert _index_vs_oracle(dump, os.path.join(tmp, "smp_o"), 31, ["--fast"]) > 10  # and the independent oracle agrees on that file


@pytest.mark.gpu
def test_gpu_index_from_reads_sampled_on_the_fly(tmp_path):
    """the same with the k-mers counted on the device from pair ranges generated by the tool's threads"""
    tmp = str(tmp_path)
    spec = _sample_spec(tmp, "smpg")
    dump = os.path.join(tmp, "smpg.dump.fq")
    subprocess.check_call([os.path.join(BIN, "rtk_build_index"), "-s", spec, "--dump-input", dump])
    want = _build(dump, os.path.join(tmp, "smpg_file"), 31, [])
    os.environ["RTK_INDEX_CAP"] = "3000000"  # several partitions of the k-mer space: the source is generated once per partition
    try:
        got = _build(spec, os.path.join(tmp, "smpg_src"), 31, ["--gpu"])
    finally:
        del os.environ["RTK_INDEX_CAP"]
    assert got[0] == want[0] and got[1] == want[1]


def test_fast_index_build_writes_the_same_files(tmp_path):
    tmp = str(tmp_path)
    for name, args in SETS:
        _same(tmp, name, _simulated(tmp, name, args), "--fast")
    # second-pass index (--colour-reads: every read its own colour) through the same fast path
    sr = _simulated(tmp, "p2", SETS[0][1])
    lr = os.path.join(tmp, "p2.lr.fq")
    outs = []
    for mode in ([], ["--fast"]):
        out = os.path.join(tmp, "p2_" + ("fast" if mode else "plain"))
        subprocess.check_call([os.path.join(BIN, "rtk_build_index"), "-s", sr, "--colour-reads", lr, "-o", out] + mode, stderr=subprocess.DEVNULL)
        outs.append((open(out + ".index.k31.fasta.gz", "rb").read(), open(out + ".index.k31.rtsk", "rb").read()))
    assert outs[0] == outs[1]
    trace = _same(tmp, "self", _self_meeting_genomes(tmp), "--fast", ks=(31,))
    n_plain = [int(l.split(":")[1].split()[0]) for l in trace.splitlines() if "chains that meet themselves" in l]
    assert n_plain and n_plain[0] >= 2, trace  # the loop and the hairpin went through the plain construction inside the fast path


def test_index_from_gzip_input_of_several_members(tmp_path):
    """Short reads (and the colour reads of a second-pass index) as ONE gzip stream and as a concatenation of gzip members (inflated member by
    member on the tool's threads, common/mgzip.hpp): the files of the plain-text input; a cut-short gzip file is refused."""
    import gzip
    tmp = str(tmp_path)
    sr = _simulated(tmp, "gz", SETS[0][1])
    lr = os.path.join(tmp, "gz.lr.fq")
    text, ltext = open(sr, "rb").read(), open(lr, "rb").read()
    def members(t, n):
        step = len(t) // n + 1
        return b"".join(gzip.compress(t[i:i + step], 1) for i in range(0, len(t), step))
    open(sr + ".one.gz", "wb").write(gzip.compress(text, 1))
    open(sr + ".many.gz", "wb").write(members(text, 9))
    open(lr + ".many.gz", "wb").write(members(ltext, 5))
    def run(s_in, l_in, out, mode):
        r = subprocess.run([os.path.join(BIN, "rtk_build_index"), "-s", s_in, "--colour-reads", l_in, "-o", out] + mode, capture_output=True, text=True)
        return r, (open(out + ".index.k31.fasta.gz", "rb").read(), open(out + ".index.k31.rtsk", "rb").read()) if r.returncode == 0 else None
    _, want = run(sr, lr, os.path.join(tmp, "plain"), [])
    for s_in, l_in, mode in ((sr + ".one.gz", lr, []), (sr + ".many.gz", lr + ".many.gz", []), (sr + ".many.gz", lr + ".many.gz", ["--fast"]), (sr + ".one.gz", lr + ".many.gz", ["--fast"])):
        r, got = run(s_in, l_in, os.path.join(tmp, "o"), mode)
        assert r.returncode == 0, r.stderr
        assert got == want, (s_in, l_in, mode)
    cut = open(lr + ".many.gz", "rb").read()
    open(lr + ".cut.gz", "wb").write(cut[:len(cut) * 2 // 3])
    r, _ = run(sr, lr + ".cut.gz", os.path.join(tmp, "bad"), ["--fast"])
    assert r.returncode != 0 and "gzip" in r.stderr
    cut = open(sr + ".one.gz", "rb").read()
    open(sr + ".cut.gz", "wb").write(cut[:len(cut) // 2])
    for mode in ([], ["--fast"]):
        r, _ = run(sr + ".cut.gz", lr, os.path.join(tmp, "bad2"), mode)
        assert r.returncode != 0 and "gzip" in r.stderr, (mode, r.stderr)


@pytest.mark.gpu
def test_gpu_index_build_writes_the_same_files(tmp_path):
    """k-mers counted on the device: the files of three seeded sets (and the self-meeting genomes) are the plain tool's, byte for byte."""
    tmp = str(tmp_path)
    for name, args in SETS[:3]: # (the fourth set exercises host code that --gpu shares with --fast: covered by the CPU tier)
        _same(tmp, name, _simulated(tmp, name, args), "--gpu")
    _same(tmp, "self", _self_meeting_genomes(tmp), "--gpu", ks=(31,))
    # a bigger set, several partitions of the k-mer space forced (RTK_INDEX_CAP: k-mers per pass)
    sr = _simulated(tmp, "c5m", ["--seed", "2", "--ref-len", "3000000", "--het", "0.001", "--sr-cov", "30", "--sr-err", "0.005"])
    a = _build(sr, os.path.join(tmp, "c5m_plain"), 31, [])
    b = _build(sr, os.path.join(tmp, "c5m_gpu"), 31, ["--gpu"])
    os.environ["RTK_INDEX_CAP"] = "30000000"
    try:
        c = _build(sr, os.path.join(tmp, "c5m_gpu_parts"), 31, ["--gpu"])
    finally:
        del os.environ["RTK_INDEX_CAP"]
    assert a[0] == b[0] == c[0] and a[1] == b[1] == c[1]


@pytest.mark.gpu
def test_gpu_index_counts_the_windows_across_the_pieces_of_a_long_record(tmp_path):
    """`-s` with records much longer than a staging buffer of the device count (RTK_INDEX_CHUNK: 4 kb here, 256 MB in production): a record is handed
    to the k-mer kernel in pieces, and the k - 1 windows across each cut must be counted (pieces overlap by k - 1 characters). Three copies of every
    record with the cuts at different offsets: a window lost at a cut would fall from 3 to 2 occurrences, below --min-count 3."""
    import random
    tmp = str(tmp_path); rnd = random.Random(77)
    recs = ["".join(rnd.choice("ACGT") for _ in range(20000 + 37 * i)) for i in range(5)]
    fa = os.path.join(tmp, "long.fa")
    with open(fa, "w") as f:
        for c in range(3):
            for i, r in enumerate(recs):
                f.write(">c%d_r%d\n%s\n" % (c, i, r if c != 1 else _rc(r)))
    def build(out, extra, env):
        r = subprocess.run([os.path.join(BIN, "rtk_build_index"), "-s", fa, "-o", out, "-k", "31", "--min-count", "3"] + extra, capture_output=True, text=True, env=dict(os.environ, RTK_INDEX_TRACE="1", **env))
        assert r.returncode == 0, r.stderr
        return open(out + ".index.k31.fasta.gz", "rb").read(), open(out + ".index.k31.rtsk", "rb").read()
    a = build(os.path.join(tmp, "plain"), [], {})
    b = build(os.path.join(tmp, "gpu"), ["--gpu"], {})
    c = build(os.path.join(tmp, "gpu_cut"), ["--gpu"], {"RTK_INDEX_CHUNK": "4096"})
    assert a == b == c
    import gzip
    seqs = [l for l in gzip.decompress(a[0]).decode().split("\n") if l and l[0] != ">"]
    assert sorted(len(x) for x in seqs) == sorted(len(r) for r in recs)  # every record came back as one unitig


@pytest.mark.gpu
def test_gpu_index_build_from_gzip_input(tmp_path):
    """`--gpu` with the short reads as gzip of several members, as one member and cut short: the device counts the k-mers of text that the host
    reader inflates (common/mgzip.hpp on the tool's threads); same files as from the plain text, the damaged file refused."""
    import gzip
    tmp = str(tmp_path)
    sr = _simulated(tmp, "gzg", SETS[0][1])
    text = open(sr, "rb").read()
    step = len(text) // 7 + 1
    open(sr + ".many.gz", "wb").write(b"".join(gzip.compress(text[i:i + step], 1) for i in range(0, len(text), step)))
    open(sr + ".one.gz", "wb").write(gzip.compress(text, 1))
    want = _build(sr, os.path.join(tmp, "plain"), 31, [])
    for inp in (sr + ".many.gz", sr + ".one.gz"):
        got = _build(inp, os.path.join(tmp, "g"), 31, ["--gpu"])
        assert got[0] == want[0] and got[1] == want[1], inp
    cut = open(sr + ".one.gz", "rb").read()
    open(sr + ".cut.gz", "wb").write(cut[:len(cut) // 2])
    r = subprocess.run([os.path.join(BIN, "rtk_build_index"), "-s", sr + ".cut.gz", "-o", os.path.join(tmp, "bad"), "--gpu"], capture_output=True, text=True)
    assert r.returncode != 0 and "gzip" in r.stderr, r.stderr


@pytest.mark.gpu
def test_gpu_index_colours_with_a_small_event_buffer(tmp_path, monkeypatch):
    """the device keeps the (unitig, read) events in a buffer that it sorts and thins out when it is half full: a buffer far smaller than the events of
    the input (many compactions) gives the same files; one that cannot hold the events between two looks is an error, not a loss"""
    tmp = str(tmp_path)
    name, args = SETS[0]
    sr = _simulated(tmp, name, args)
    a = _build(sr, os.path.join(tmp, "ev_plain"), 31, [])
    import re
    monkeypatch.setenv("RTK_INDEX_THREADS", "16")
    b = _build(sr, os.path.join(tmp, "ev_gpu0"), 31, ["--gpu"])  # (default buffer: how many distinct events the set has)
    n_distinct = int(re.search(r"-> (\d+) distinct", b[2]).group(1))
    monkeypatch.setenv("RTK_INDEX_EVENTS", str(n_distinct * 3 // 2))  # half of it is less than the distinct events alone: thinned out while the reads still come
    b = _build(sr, os.path.join(tmp, "ev_gpu"), 31, ["--gpu"])
    assert a[0] == b[0] and a[1] == b[1]
    assert int(re.search(r"thinned out (\d+) times", b[2]).group(1)) >= 1, b[2]  # (how often depends on how many chunks the reader cuts the set into: at least once before the end)
    monkeypatch.setenv("RTK_INDEX_EVENTS", str(max(1024, n_distinct // 4)))
    r = subprocess.run([os.path.join(BIN, "rtk_build_index"), "-s", sr, "-o", os.path.join(tmp, "ev_bad"), "--gpu"], capture_output=True, text=True)
    assert r.returncode != 0 and "events" in r.stderr, r.stderr

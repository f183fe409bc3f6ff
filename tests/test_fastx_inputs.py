"""Input layouts of the long-read files (`-l`, `-L`): the reference reads them with Bifrost's FileParser = kseq (src/Ratatosk.cpp:658,767):
FASTA or FASTQ, sequence and quality on any number of lines, name = header up to the first whitespace, CR LF tolerated, gzip transparent.
The host driver has two readers (byte ranges on `-c` threads for plain / blocked-gzip files on four lines; one thread for the rest); every
layout of the same reads must give the same output file, and a layout the byte-range reader cannot take must be refused, not misread.
CPU tier: the C++ driver linked against the simulator."""
import gzip
import hashlib
import os
import random
import subprocess

import pytest

from conftest import BIN, ROOT, make_dataset
from oracle import oracle_py as op

SIM = os.path.join(ROOT, "tests", "hostsim", "Ratatosk_sim")


def _wrap(s, w):
    return "\n".join(s[i:i + w] for i in range(0, len(s), w)) if s else ""


@pytest.fixture(scope="module")
def ds(tmp_path_factory):
    return make_dataset(tmp_path_factory.mktemp("ds_layouts"), "lay", ["--seed", 3, "--ref-len", 20000, "--sr-cov", 30, "--sr-err", 0.003, "--lr-n", 8, "--lr-len", 1500,
                                                                       "--lr-profile", "ont", "--lr-err", 0.08])


def _run(ds, reads, out, env=None, extra=()):
    e = dict(os.environ, RTK_SIM_DEVICES="1")
    e.update(env or {})
    return subprocess.run([SIM, "correct", "-1", "-c", "3", "-B", "4000", "-g", ds + ".index.k31.fasta.gz", "-d", ds + ".index.k31.rtsk", "-l", reads, "-o", out] + list(extra),
                          capture_output=True, text=True, env=e)


def _sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def test_every_layout_of_the_same_reads_gives_the_same_file(ds, tmp_path):
    recs = op.read_fastq(ds + ".lr.fq")
    r = _run(ds, ds + ".lr.fq", str(tmp_path / "base"))
    assert r.returncode == 0, r.stderr
    want = _sha(str(tmp_path / "base.2.fastq"))
    # ... which is what the oracle says
    og = op.Graph(ds + ".index.k31.fasta.gz", ds + ".index.k31.rtsk", 31)
    out, _ = og.correct_batch([x[1] for x in recs], [x[2] for x in recs], threads=4)
    assert [(g[1], g[2]) for g in op.read_fastq(str(tmp_path / "base.2.fastq"))] == out
    rng = random.Random(5)
    nasty = ["".join(rng.choice("@+>I5") for _ in x[1]) for x in recs]  # quality lines that begin like headers and separators
    layouts = {
        "crlf.fq": "".join("@%s\r\n%s\r\n+\r\n%s\r\n" % x for x in recs),
        "multi.fa": "".join(">%s\n%s\n" % (x[0], _wrap(x[1], 70)) for x in recs),
        "one_line.fa": "".join(">%s some comment\n%s\n" % (x[0], x[1]) for x in recs),
        "comment.fq": "".join("@%s comment\tx=1\n%s\n+%s comment\n%s\n" % (x[0], x[1], x[0], x[2]) for x in recs),
        "multi.fq": "".join("@%s\n%s\n+\n%s\n" % (x[0], _wrap(x[1], 61), _wrap(q, 61)) for x, q in zip(recs, nasty)),
        "nasty4.fq": "".join("@%s\n%s\n+\n%s\n" % (x[0], x[1], q) for x, q in zip(recs, nasty)),
        "blank_lines.fq": "".join("@%s\n%s\n+\n%s\n\n" % x for x in recs),
        "no_final_newline.fq": "".join("@%s\n%s\n+\n%s\n" % x for x in recs)[:-1],
        "lower.fq": "".join("@%s\n%s\n+\n%s\n" % (x[0], x[1].lower(), x[2]) for x in recs),
    }
    for name, text in layouts.items():
        p = str(tmp_path / name)
        open(p, "w", newline="").write(text)
        for env in ({}, {"RTK_SERIAL_READER": "1"}):
            r = _run(ds, p, str(tmp_path / "o"), env)
            assert r.returncode == 0, (name, r.stderr)
            assert _sha(str(tmp_path / "o.2.fastq")) == want, (name, env)
    # compressed: an ordinary gzip stream (one-thread reader) and blocked gzip (byte ranges), four lines and several
    for name in ("multi.fq", "nasty4.fq", "multi.fa"):
        p = str(tmp_path / name)
        with gzip.open(p + ".gz", "wb") as f:
            f.write(open(p, "rb").read())
        subprocess.check_call([os.path.join(BIN, "rtk_bgzip"), p, p + ".bgz.gz"])
        for q in (p + ".gz", p + ".bgz.gz"):
            r = _run(ds, q, str(tmp_path / "o"))
            assert r.returncode == 0, (q, r.stderr)
            assert _sha(str(tmp_path / "o.2.fastq")) == want, q


def test_a_layout_the_range_reader_cannot_take_is_refused_not_misread(ds, tmp_path):
    recs = op.read_fastq(ds + ".lr.fq")
    r = _run(ds, ds + ".lr.fq", str(tmp_path / "base"))
    want = _sha(str(tmp_path / "base.2.fastq"))
    # four lines first, several lines further down: the sniff at the head of the file cannot see it
    p = str(tmp_path / "mixed.fq")
    open(p, "w").write("@%s\n%s\n+\n%s\n" % recs[0] + "".join("@%s\n%s\n+\n%s\n" % (x[0], _wrap(x[1], 80), _wrap(x[2], 80)) for x in recs[1:]))
    r = _run(ds, p, str(tmp_path / "bad"))
    assert r.returncode != 0 and "laid out differently" in r.stderr and not os.path.exists(str(tmp_path / "bad.2.fastq"))
    # the reader-alone mode reports the same file as a failure, not a rate over the part it could parse
    rp = subprocess.run([os.path.join(ROOT, "tests", "hostsim", "Ratatosk_sim"), "correct", "-1", "--parse-only", "-c", "3", "-B", "500", "-l", p], capture_output=True, text=True)
    assert rp.returncode != 0 and "not laid out as 4-line FASTQ" in rp.stderr and "bases/s" not in rp.stdout, rp.stdout + rp.stderr
    r = _run(ds, p, str(tmp_path / "ok"), {"RTK_SERIAL_READER": "1"})
    assert r.returncode == 0 and _sha(str(tmp_path / "ok.2.fastq")) == want
    # no reads at all
    open(str(tmp_path / "empty.fq"), "w").close()
    r = _run(ds, str(tmp_path / "empty.fq"), str(tmp_path / "e"))
    assert r.returncode == 0 and os.path.getsize(str(tmp_path / "e.2.fastq")) == 0
    # an empty read and a read shorter than k between ordinary ones: kept, in place (src/Ratatosk.cpp:808-864 corrects what it is given)
    p = str(tmp_path / "short.fq")
    open(p, "w").write("@%s\n%s\n+\n%s\n" % recs[0] + "@empty\n\n+\n\n@short\nACGTACGT\n+\nIIIIIIII\n" + "@%s\n%s\n+\n%s\n" % recs[1])
    r = _run(ds, p, str(tmp_path / "s"))
    assert r.returncode == 0, r.stderr
    got = op.read_fastq(str(tmp_path / "s.2.fastq"))
    assert [g[0] for g in got] == [recs[0][0], "empty", "short", recs[1][0]] and got[1][1] == "" and got[2][1] == "ACGTACGT"


def test_second_pass_reads_qualities_on_several_lines(ds, tmp_path):
    """`-2` carries the input qualities (src/Correction.cpp:779,808,941): the pass-1 file and the uncorrected file re-written on several
    lines (and the uncorrected one as FASTA) must give the output of the four-line files."""
    recs = op.read_fastq(ds + ".lr.fq")
    og = op.Graph(ds + ".index.k31.fasta.gz", ds + ".index.k31.rtsk", 31)
    out, _ = og.correct_batch([x[1] for x in recs], [x[2] for x in recs], threads=4)
    p1 = str(tmp_path / "pass1.fq")
    open(p1, "w").write("".join("@%s\n%s\n+\n%s\n" % (x[0], s, q) for x, (s, q) in zip(recs, out)))
    subprocess.check_call([os.path.join(BIN, "rtk_build_index"), "-s", ds + ".sr.fq", "--colour-reads", p1, "-o", str(tmp_path / "p2")], stderr=subprocess.DEVNULL)
    p1m, rawm, rawfa = str(tmp_path / "pass1_multi.fq"), str(tmp_path / "raw_multi.fq"), str(tmp_path / "raw.fa")
    open(p1m, "w").write("".join("@%s\n%s\n+\n%s\n" % (x[0], _wrap(s, 53), _wrap(q, 53)) for x, (s, q) in zip(recs, out)))
    open(rawm, "w").write("".join("@%s\n%s\n+\n%s\n" % (x[0], _wrap(x[1], 77), _wrap(x[2], 77)) for x in recs))
    open(rawfa, "w").write("".join(">%s\n%s\n" % (x[0], _wrap(x[1], 60)) for x in recs))
    env = dict(os.environ, RTK_SIM_DEVICES="1")
    shas = []
    p1gz, rawgz = p1m + ".gz", rawm + ".gz"  # both files as gzip of several members: two member readers side by side
    for src, dst in ((p1m, p1gz), (rawm, rawgz)):
        t = open(src, "rb").read()
        open(dst, "wb").write(b"".join(gzip.compress(t[a:a + 20011], 1) for a in range(0, len(t), 20011)))
    for l, L in ((p1, ds + ".lr.fq"), (p1m, rawm), (p1m, rawfa), (p1, rawfa), (p1gz, rawgz), (p1gz, rawfa)):
        o = str(tmp_path / "o2")
        r = subprocess.run([SIM, "correct", "-2", "-K", "31", "-c", "4", "-B", "5000", "-g", str(tmp_path / "p2.index.k31.fasta.gz"), "-d", str(tmp_path / "p2.index.k31.rtsk"),
                            "-l", l, "-L", L, "-o", o], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        shas.append(_sha(o + ".fastq"))
    assert len(set(shas)) == 1
    want = og2 = op.Graph(str(tmp_path / "p2.index.k31.fasta.gz"), str(tmp_path / "p2.index.k31.rtsk"), 31)
    exp = og2.correct_batch2([s for s, _ in out], [q for _, q in out], [x[1] for x in recs], og2.opts(long_read_correct=1), threads=4)
    assert [(g[1], g[2]) for g in op.read_fastq(str(tmp_path / "o2.fastq"))] == exp


def test_range_reader_against_the_one_thread_reader_on_random_files(ds, tmp_path):
    """Random 4-line FASTQ files of reads shorter than k (nothing to correct: the output is the parse), quality lines full of '@', '+' and
    '>' — first characters included, the LAST record's too: the file end is where the record-start rule (a '@' line with a '+' line two
    below) has the fewest lines to look at. Byte ranges of 256 bytes to a few KB against the one-thread reader."""
    rng = random.Random(99)
    for it in range(24):
        n = rng.randint(1, 40)
        recs = []
        for i in range(n):
            L = rng.choice([0, 1, 2, 5, 12, 25, 30]) if rng.random() < 0.7 else rng.randint(0, 30)
            recs.append(("r%d_%d" % (it, i), "".join(rng.choice("ACGT") for _ in range(L)), "".join(rng.choice("@+>@+I") for _ in range(L))))
        text = "".join("@%s\n%s\n+\n%s\n" % x for x in recs)
        if it % 3 == 1:
            text = text[:-1]  # no newline at the end
        if it % 3 == 2:
            text = text.replace("\n", "\r\n")
        p = str(tmp_path / ("f%d.fq" % it))
        open(p, "w", newline="").write(text)
        outs = []
        for env, B in (({"RTK_SERIAL_READER": "1"}, 1 << 20), ({}, 1), ({}, rng.randint(300, 3000)), ({}, 1 << 20)):
            r = _run(ds, p, str(tmp_path / "o"), env, ["-B", str(B)])
            assert r.returncode == 0, r.stderr
            outs.append(open(str(tmp_path / "o.2.fastq")).read())
        got = [l for l in outs[0].split("\n")]
        assert [got[i][1:] for i in range(0, len(got) - 1, 4)] == [x[0] for x in recs], it
        assert [got[i] for i in range(1, len(got) - 1, 4)] == [x[1] for x in recs], it
        assert outs[1] == outs[0] and outs[2] == outs[0] and outs[3] == outs[0], it


def test_several_input_files_and_arguments_that_belong_to_no_option(ds, tmp_path):
    """Several `-l`, a text file of paths (src/Common.cpp:412-446), FASTQ + FASTA + gzip mixed: the reads in the order of the files.
    `-l a b` corrects `a` only, like the reference's getopt_long loop (src/Ratatosk.cpp:186-300) — but says so."""
    recs = op.read_fastq(ds + ".lr.fq")
    a, b, c = str(tmp_path / "a.fq"), str(tmp_path / "b.fa"), str(tmp_path / "c.fq.gz")
    open(a, "w").write("".join("@%s\n%s\n+\n%s\n" % x for x in recs[:3]))
    open(b, "w").write("".join(">%s\n%s\n" % (x[0], x[1]) for x in recs[3:5]))
    with gzip.open(c, "wt") as f:
        f.write("".join("@%s\n%s\n+\n%s\n" % x for x in recs[5:]))
    lst = str(tmp_path / "list.txt")
    open(lst, "w").write("%s\n%s\n%s\n" % (a, b, c))
    r = _run(ds, ds + ".lr.fq", str(tmp_path / "base"))
    want = _sha(str(tmp_path / "base.2.fastq"))
    r = _run(ds, lst, str(tmp_path / "o1"))
    assert r.returncode == 0 and _sha(str(tmp_path / "o1.2.fastq")) == want, r.stderr
    r = _run(ds, a, str(tmp_path / "o2"), extra=["-l", b, "-l", c])
    assert r.returncode == 0 and _sha(str(tmp_path / "o2.2.fastq")) == want, r.stderr
    r = _run(ds, a, str(tmp_path / "o3"), extra=[b, c])
    assert r.returncode == 0 and r.stderr.count("belongs to no option") == 2
    assert [g[0] for g in op.read_fastq(str(tmp_path / "o3.2.fastq"))] == [x[0] for x in recs[:3]]
    r = _run(ds, str(tmp_path / "missing.fq"), str(tmp_path / "o4"))
    assert r.returncode == 1 and "cannot open" in r.stderr


def test_gzip_of_several_members_inflated_on_several_threads(ds, tmp_path):
    """`cat a.fq.gz b.fq.gz ...` (one gzip member per original file) is inflated member by member on the `-c` threads (common/mgzip.hpp): member
    starts are found by their magic bytes and confirmed by walking the chain of member ends. Same output as the plain file, whatever the
    members look like: empty ones, records that straddle two members, the magic bytes inside a member's own data (stored blocks, so that
    they stand verbatim in the file), bytes behind the last member that are no member (zlib ignores them). A member that is cut short or
    damaged is an error, not the end of the input."""
    recs = op.read_fastq(ds + ".lr.fq")
    r = _run(ds, ds + ".lr.fq", str(tmp_path / "base"))
    want = _sha(str(tmp_path / "base.2.fastq"))
    magic = "\x1f\x8b\x08\x00" * 40
    text = "".join("@%s %s\n%s\n+\n%s\n" % (x[0], magic, x[1], x[2]) for x in recs).encode("latin-1")
    cuts = [0, 1, len(text) // 5, len(text) // 5 + 1, len(text) // 2, len(text) - 7, len(text)]  # pieces of any size, cut anywhere
    pieces = [text[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    blob = b"".join(gzip.compress(p, lvl) for p, lvl in zip(pieces, [6, 0, 0, 9, 0, 1])) + gzip.compress(b"")
    assert blob.count(b"\x1f\x8b\x08\x00") > len(pieces) + 50  # false member starts inside the stored blocks
    for name, data in (("members.fq.gz", blob), ("members_then_junk.fq.gz", blob + b"\x00\x00\x00junk")):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        assert gzip.decompress(blob) == text
        for cores in ("1", "2", "7"):
            rr = subprocess.run([SIM, "correct", "-1", "-c", cores, "-B", "4000", "-g", ds + ".index.k31.fasta.gz", "-d", ds + ".index.k31.rtsk", "-l", p, "-o", str(tmp_path / "o")],
                                capture_output=True, text=True, env=dict(os.environ, RTK_SIM_DEVICES="1"))
            assert rr.returncode == 0, rr.stderr
            assert _sha(str(tmp_path / "o.2.fastq")) == want, (name, cores)
    # cut short inside the third member / one byte flipped in the fourth: refused by both readers
    third = len(gzip.compress(pieces[0], 6)) + len(gzip.compress(pieces[1], 0)) + 1000
    bad1 = str(tmp_path / "cut.fq.gz")
    open(bad1, "wb").write(blob[:third])
    flipped = bytearray(blob)
    at = len(blob) - len(gzip.compress(b"")) - len(gzip.compress(pieces[5], 1)) - len(gzip.compress(pieces[4], 0)) - 200
    flipped[at] ^= 0x55
    bad2 = str(tmp_path / "flipped.fq.gz")
    open(bad2, "wb").write(bytes(flipped))
    # a last member of which fewer than 18 bytes are left (header cut), and an empty last member cut short: zlib's gzread reports an
    # unexpected end of file on both, so they are errors, not trailing junk
    whole = b"".join(gzip.compress(p, lvl) for p, lvl in zip(pieces, [6, 0, 0, 9, 0, 1]))
    bad3, bad4 = str(tmp_path / "cut_header.fq.gz"), str(tmp_path / "cut_empty.fq.gz")
    open(bad3, "wb").write(whole + gzip.compress(pieces[0], 6)[:9])
    open(bad4, "wb").write(whole + gzip.compress(b"")[:13])
    for p in (bad1, bad2, bad3, bad4):
        for cores in ("1", "4"):
            rr = subprocess.run([SIM, "correct", "-1", "-c", cores, "-g", ds + ".index.k31.fasta.gz", "-d", ds + ".index.k31.rtsk", "-l", p, "-o", str(tmp_path / "bad")],
                                capture_output=True, text=True, env=dict(os.environ, RTK_SIM_DEVICES="1"))
            assert rr.returncode != 0 and "gzip" in rr.stderr, (p, cores, rr.stderr)


def test_gzip_streams_of_every_block_kind_through_the_fast_decoder(ds, tmp_path):
    """With `-c` > 1 gzip members are decoded by common/finflate.hpp instead of zlib. The same reads compressed so that the stream consists of
    stored blocks (level 0), fixed-Huffman blocks (Z_FIXED), literal-only dynamic blocks (Z_HUFFMAN_ONLY), distance-1 matches (Z_RLE), ordinary
    blocks at levels 1 / 6 / 9 with full-flush points in between, and a small window (so that distances wrap the 32 KB history differently):
    always the plain file's output, and the same with RTK_ZLIB_INFLATE=1 (zlib inside the same reader)."""
    import zlib
    text = open(ds + ".lr.fq", "rb").read()
    r = _run(ds, ds + ".lr.fq", str(tmp_path / "base"))
    want = _sha(str(tmp_path / "base.2.fastq"))
    variants = [(0, zlib.Z_DEFAULT_STRATEGY, 15), (6, zlib.Z_FIXED, 15), (6, zlib.Z_HUFFMAN_ONLY, 15), (6, zlib.Z_RLE, 15), (1, zlib.Z_DEFAULT_STRATEGY, 15),
                (6, zlib.Z_DEFAULT_STRATEGY, 15), (9, zlib.Z_DEFAULT_STRATEGY, 15), (9, zlib.Z_DEFAULT_STRATEGY, 9), (6, zlib.Z_FILTERED, 12)]
    for i, (lvl, strat, wbits) in enumerate(variants):
        co = zlib.compressobj(lvl, zlib.DEFLATED, 16 + wbits, 9, strat)
        cut = [0, len(text) // 3, len(text) // 3 + 5, len(text)]
        blob = b"".join(co.compress(text[a:b]) + co.flush(zlib.Z_FULL_FLUSH) for a, b in zip(cut[:-1], cut[1:])) + co.flush()
        assert gzip.decompress(blob) == text
        p = str(tmp_path / ("v%d.fq.gz" % i))
        open(p, "wb").write(blob)
        for env in ({}, {"RTK_ZLIB_INFLATE": "1"}):
            rr = _run(ds, p, str(tmp_path / "o"), env)
            assert rr.returncode == 0, (i, rr.stderr)
            assert _sha(str(tmp_path / "o.2.fastq")) == want, (i, env)


def test_a_damaged_bgzf_block_is_an_error(ds, tmp_path):
    """Blocked gzip goes through the byte-range reader, whose blocks are inflated by the reader's own decoder with the block's CRC-32 checked
    (zlib's raw inflate, used before, did not look at it): one flipped byte in the middle of the file must stop the run."""
    text = open(ds + ".lr.fq", "rb").read() * 6
    p = str(tmp_path / "in.fq")
    open(p, "wb").write(text)
    subprocess.check_call([os.path.join(BIN, "rtk_bgzip"), p, p + ".gz"])
    r = _run(ds, p + ".gz", str(tmp_path / "ok"))
    assert r.returncode == 0, r.stderr
    b = bytearray(open(p + ".gz", "rb").read())
    b[len(b) // 2] ^= 0x10
    open(p + ".bad.gz", "wb").write(bytes(b))
    for env in ({}, {"RTK_ZLIB_INFLATE": "1"}):
        r = _run(ds, p + ".bad.gz", str(tmp_path / "bad"), env)
        assert r.returncode != 0 and ("read error" in r.stderr or "gzip" in r.stderr), r.stderr


def test_range_reader_on_random_multi_line_fasta(ds, tmp_path):
    """The byte-range reader takes FASTA on any number of lines (a record starts at every line that begins with '>'): random files of short
    records with empty lines, empty sequences and lines of every length, ranges from 256 bytes up, against the one-thread reader."""
    rng = random.Random(123)
    for it in range(16):
        n = rng.randint(1, 50)
        recs, text = [], []
        for i in range(n):
            L = rng.choice([0, 1, 3, 17, 29, 30])
            s = "".join(rng.choice("ACGT") for _ in range(L))
            w = rng.choice([1, 2, 7, 60])
            body = "".join(s[a:a + w] + "\n" + ("\n" if rng.random() < 0.1 else "") for a in range(0, len(s), w)) if s else ("\n" if rng.random() < 0.5 else "")
            text.append(">r%d_%d some words\n%s" % (it, i, body))
            recs.append(("r%d_%d" % (it, i), s))
        t = "".join(text)
        if it % 4 == 1:
            t = t.rstrip("\n")
        if it % 4 == 2:
            t = t.replace("\n", "\r\n")
        p = str(tmp_path / ("f%d.fa" % it))
        open(p, "w", newline="").write(t)
        outs = []
        for env, B in (({"RTK_SERIAL_READER": "1"}, 1 << 20), ({}, 1), ({}, rng.randint(300, 3000)), ({}, 1 << 20)):
            r = _run(ds, p, str(tmp_path / "o"), env, ["-B", str(B)])
            assert r.returncode == 0, r.stderr
            outs.append(open(str(tmp_path / "o.2.fastq")).read())
        got = outs[0].split("\n")
        assert [got[i][1:] for i in range(0, len(got) - 1, 4)] == [x[0] for x in recs], it
        assert [got[i] for i in range(1, len(got) - 1, 4)] == [x[1] for x in recs], it
        assert outs[1] == outs[0] and outs[2] == outs[0] and outs[3] == outs[0], it

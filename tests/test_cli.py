"""`Ratatosk correct -1` CLI surface of the build (reference: src/Ratatosk.cpp:145-301,1029-1037): option handling without a GPU,
and on the GPU the written OUT.2.fastq must equal the oracle's records, in input order."""
import os
import subprocess

import pytest

from conftest import BIN, ROOT
from oracle import oracle_py as op

EXE = os.path.join(BIN, "Ratatosk")


def test_cli_usage_and_scope_messages():
    assert subprocess.run([EXE, "--help"], capture_output=True).returncode == 0
    r = subprocess.run([EXE, "index", "-s", "x"], capture_output=True, text=True)
    assert r.returncode == 1 and "not in scope" in r.stderr
    r = subprocess.run([EXE, "correct", "-2", "-g", "a", "-d", "b", "-l", "c", "-o", "d"], capture_output=True, text=True)
    assert r.returncode == 0 and "-L" in r.stderr  # the second pass needs the uncorrected reads as well
    r = subprocess.run([EXE, "correct", "-g", "a", "-d", "b", "-l", "c", "-o", "d"], capture_output=True, text=True)
    assert r.returncode == 1 and "-1 or -2" in r.stderr  # both passes in one run would need the index step, which is out of scope


def test_cli_cores_keep_the_reference_meaning(ds_small, tmp_path):
    """-c is the number of host threads and must not exceed the hardware concurrency (src/Ratatosk.cpp:312-322: message, return 0)."""
    r = subprocess.run([EXE, "correct", "-1", "-c", "100000", "-g", "a", "-d", "b", "-l", "c", "-o", "d"], capture_output=True, text=True)
    assert r.returncode == 0 and "Number of threads cannot be greater" in r.stderr
    r = subprocess.run([EXE, "correct", "-1", "-c", "0", "-g", "a", "-d", "b", "-l", "c", "-o", "d"], capture_output=True, text=True)
    assert r.returncode == 0 and "less than or equal to 0" in r.stderr


def test_cli_partial_output_is_removed_on_failure(ds_small, tmp_path):
    """A ticket that fails (here: an input file that does not exist, second in the list) must not leave an OUT.2.fastq that silently
    lacks reads (simulator build of the driver)."""
    import os as _os
    sim = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "hostsim", "Ratatosk_sim")
    lst = str(tmp_path / "inputs.txt")
    with open(lst, "w") as f:
        f.write(ds_small + ".lr.fq\n" + str(tmp_path / "missing.fq") + "\n")
    out = str(tmp_path / "out")
    r = subprocess.run([sim, "correct", "-1", "-c", "1", "-B", "6000", "-g", ds_small + ".index.k31.fasta.gz", "-d", ds_small + ".index.k31.rtsk", "-l", lst, "-o", out], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open input file" in r.stderr
    assert not _os.path.exists(out + ".2.fastq")


def test_cli_fails_loudly_without_gpu(ds_small, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([EXE, "correct", "-1", "-g", ds_small + ".index.k31.fasta.gz", "-d", ds_small + ".index.k31.rtsk", "-l", ds_small + ".lr.fq",
                        "-o", str(tmp_path / "out")], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_cli_correct_matches_oracle(ds_small, tmp_path):
    fa, rt = ds_small + ".index.k31.fasta.gz", ds_small + ".index.k31.rtsk"
    out = str(tmp_path / "out")
    # small batches so that the ticket reorder path is exercised
    r = subprocess.run([EXE, "correct", "-1", "-v", "-c", "1", "-B", "12000", "-g", fa, "-d", rt, "-l", ds_small + ".lr.fq", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = op.read_fastq(out + ".2.fastq")
    reads = op.read_fastq(ds_small + ".lr.fq")
    want, _ = op.Graph(fa, rt, 31).correct_batch([x[1] for x in reads], [x[2] for x in reads], threads=4)
    assert [g[0] for g in got] == [x[0] for x in reads]
    assert [(g[1], g[2]) for g in got] == want


@pytest.mark.gpu
def test_cli_two_workers_many_tickets(ds_medium, tmp_path):
    """Seven tickets through the two overlapping workers of one GPU (seed stage of one, region stage of the other): records still
    come out in input order and equal the oracle's."""
    fa, rt = ds_medium + ".index.k31.fasta.gz", ds_medium + ".index.k31.rtsk"
    out = str(tmp_path / "out")
    r = subprocess.run([EXE, "correct", "-1", "-c", "1", "-B", "200000", "-g", fa, "-d", rt, "-l", ds_medium + ".lr.fq", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = op.read_fastq(out + ".2.fastq")
    reads = op.read_fastq(ds_medium + ".lr.fq")
    want, _ = op.Graph(fa, rt, 31).correct_batch([x[1] for x in reads], [x[2] for x in reads], threads=os.cpu_count() or 4)
    assert [g[0] for g in got] == [x[0] for x in reads]
    assert [(g[1], g[2]) for g in got] == want


@pytest.mark.gpu
def test_cli_fasta_input_through_a_list_file(ds_small, tmp_path):
    """Inputs may be FASTA (no qualities; pass 1 writes synthetic ones anyway, src/Correction.cpp:184-185) and may be given as a text
    file listing one path per line (src/Common.cpp:412-446), here two files whose reads must come out concatenated in order."""
    fa, rt = ds_small + ".index.k31.fasta.gz", ds_small + ".index.k31.rtsk"
    reads = op.read_fastq(ds_small + ".lr.fq")
    f1, f2, lst = str(tmp_path / "a.fa"), str(tmp_path / "b.fasta"), str(tmp_path / "inputs.txt")
    half = len(reads) // 2
    with open(f1, "w") as f:
        for n, s, _ in reads[:half]:
            f.write(">%s some description\n" % n)
            for i in range(0, len(s), 70):  # multi-line FASTA
                f.write(s[i:i + 70] + "\n")
    with open(f2, "w") as f:
        for n, s, _ in reads[half:]:
            f.write(">%s\n%s\n" % (n, s.lower()))
    with open(lst, "w") as f:
        f.write(f1 + "\n" + f2 + "\n")
    out = str(tmp_path / "out")
    r = subprocess.run([EXE, "correct", "-1", "-c", "1", "-B", "9000", "-g", fa, "-d", rt, "-l", lst, "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = op.read_fastq(out + ".2.fastq")
    want, _ = op.Graph(fa, rt, 31).correct_batch([x[1] for x in reads], None, threads=4)
    assert [g[0] for g in got] == [x[0] for x in reads]
    assert [(g[1], g[2]) for g in got] == want


@pytest.mark.gpu
def test_cli_snp_annotated_index_and_min_conf_option(ds_snps, tmp_path):
    """-m / --min-conf-snp-corr (src/Ratatosk.cpp:240) reaches fixAmbiguity; --strip-annotations gives the plain index."""
    fa, rt = ds_snps + ".index.k31.fasta.gz", ds_snps + ".index.k31.rtsk"
    reads = op.read_fastq(ds_snps + ".lr.fq")
    seqs, quals = [x[1] for x in reads], [x[2] for x in reads]
    out = str(tmp_path / "out")
    r = subprocess.run([EXE, "correct", "-1", "-c", "1", "-m", "0.5", "-g", fa, "-d", rt, "-l", ds_snps + ".lr.fq", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    og = op.Graph(fa, rt, 31)
    want, _ = og.correct_batch(seqs, quals, opts=og.opts(min_confidence_snp_corr=0.5), threads=4)
    assert [(g[1], g[2]) for g in op.read_fastq(out + ".2.fastq")] == want
    r = subprocess.run([EXE, "correct", "-1", "-c", "1", "--strip-annotations", "-g", fa, "-d", rt, "-l", ds_snps + ".lr.fq", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0 and "dropped" in r.stderr, r.stderr
    plain, _ = op.Graph(ds_snps + "_plain.index.k31.fasta.gz", ds_snps + "_plain.index.k31.rtsk", 31).correct_batch(seqs, quals, threads=4)
    assert [(g[1], g[2]) for g in op.read_fastq(out + ".2.fastq")] == plain
    assert want != plain


def _oracle_rounds(og, seqs, quals, n_rounds):
    """The round loop of src/Ratatosk.cpp:847-866 around the oracle's per-read correction."""
    base = og.opts()
    f0, w0 = base.weak_region_len_factor, base.max_len_weak_region1
    for j in range(n_rounds):
        o = og.opts()
        o.min_score = 1.0 - (j + 1) * (1.0 / n_rounds)
        o.weak_region_len_factor = f0 - (n_rounds - j - 1) * ((f0 - 0.10) / (n_rounds - 1))
        o.max_len_weak_region1 = (j + 1) * (w0 // n_rounds)
        out, _ = og.correct_batch(seqs, quals, opts=o, threads=4)
        seqs, quals = [s for s, _ in out], [q for _, q in out]
    return list(zip(seqs, quals))


def test_cli_correction_rounds_and_reference_flags(ds_small, tmp_path):
    """-r N (src/Ratatosk.cpp:847-866: N passes over a read with min_score / length window / maximum region length stepping towards
    the one-round values) through the simulator build of the driver against the oracle; the flags a reference command line may carry
    that only `index` reads (-F -O -I -S -M -C) are accepted, the out-of-scope inputs (-u -a -p -P -f) are named and refused."""
    sim = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "Ratatosk_sim")
    fa, rt = ds_small + ".index.k31.fasta.gz", ds_small + ".index.k31.rtsk"
    og = op.Graph(fa, rt, 31)
    reads = op.read_fastq(ds_small + ".lr.fq")
    want = _oracle_rounds(og, [r[1] for r in reads], [r[2] for r in reads], 3)
    assert want != og.correct_batch([r[1] for r in reads], [r[2] for r in reads], threads=4)[0]  # the rounds make a difference on this set
    out = str(tmp_path / "r3")
    r = subprocess.run([sim, "correct", "-1", "-r", "3", "-F", "-O", "-I", "-S", "0.5", "-c", "2", "-B", "9000", "-g", fa, "-d", rt, "-l", ds_small + ".lr.fq", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = op.read_fastq(out + ".2.fastq")
    assert [(g[1], g[2]) for g in got] == want and [g[0] for g in got] == [x[0] for x in reads]
    r = subprocess.run([sim, "correct", "-1", "-u", "x.fq", "-g", fa, "-d", rt, "-l", ds_small + ".lr.fq", "-o", out], capture_output=True, text=True)
    assert r.returncode == 1 and "not in scope" in r.stderr
    r = subprocess.run([sim, "correct", "-1", "-r", "0", "-g", fa, "-d", rt, "-l", ds_small + ".lr.fq", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0 and "correction rounds" in r.stderr


@pytest.mark.gpu
def test_gpu_cli_correction_rounds(ds_small, tmp_path):
    fa, rt = ds_small + ".index.k31.fasta.gz", ds_small + ".index.k31.rtsk"
    og = op.Graph(fa, rt, 31)
    reads = op.read_fastq(ds_small + ".lr.fq")
    want = _oracle_rounds(og, [r[1] for r in reads], [r[2] for r in reads], 2)
    out = str(tmp_path / "r2")
    r = subprocess.run([EXE, "correct", "-1", "-r", "2", "-c", "2", "-B", "9000", "-g", fa, "-d", rt, "-l", ds_small + ".lr.fq", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert [(g[1], g[2]) for g in op.read_fastq(out + ".2.fastq")] == want


def test_parallel_reader_cuts_files_like_the_serial_one(ds_small, tmp_path):
    """First pass on plain files: -c threads parse byte ranges of the input independently (rtk::PlainChunks). With ranges of a few hundred
    bytes every kind of boundary occurs: records longer than a range, quality lines that start with '@' or '+', CRLF line ends, a FASTA file
    with multi-line records next to the FASTQ one. The output must be the serial reader's, byte for byte (simulator build of the driver)."""
    import random
    rnd = random.Random(4)
    reads = op.read_fastq(ds_small + ".lr.fq")[:6]
    fq, fa = str(tmp_path / "in.fq"), str(tmp_path / "in.fa")
    with open(fq, "w", newline="") as f:
        for i, (name, s, q) in enumerate(reads):
            lead = "@+I5"[i % 4]
            qq = lead + "".join(rnd.choice("@+5I!") for _ in range(len(s) - 1))
            eol = "\r\n" if i % 2 else "\n"
            f.write("@%s some comment%s%s%s+%s%s%s" % (name, eol, s, eol, eol, qq, eol))
    with open(fa, "w") as f:
        for name, s, q in reads[:3]:
            f.write(">%s_fa\n" % name + "\n".join(s[i:i + 70] for i in range(0, len(s), 70)) + "\n")
    lst = str(tmp_path / "list.txt")
    open(lst, "w").write(fq + "\n" + fa + "\n")
    exe = os.path.join(ROOT, "tests", "hostsim", "Ratatosk_sim")
    outs = []
    for tag, env, batch in (("serial", dict(os.environ, RTK_SERIAL_READER="1"), "100000"), ("ranges_of_256_bytes", dict(os.environ), "100"), ("ranges_of_one_ticket", dict(os.environ), "3000")):
        out = str(tmp_path / tag)
        r = subprocess.run([exe, "correct", "-1", "-c", "4", "-B", batch, "-g", ds_small + ".index.k31.fasta.gz", "-d", ds_small + ".index.k31.rtsk", "-l", lst, "-o", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(open(out + ".2.fastq", "rb").read())
    assert outs[0] == outs[1] == outs[2] and outs[0].count(b"\n") == 4 * 9
    r = subprocess.run([exe, "correct", "-1", "--parse-only", "-c", "3", "-B", "100", "-l", lst], capture_output=True, text=True)
    assert r.returncode == 0 and "9 reads" in r.stdout, r.stdout + r.stderr


def _bgzf_blocks(data, block=700):
    """BGZF written by an independent encoder (python zlib): gzip members with the 'BC' extra field, `block` text bytes each, + the EOF block."""
    import struct
    import zlib
    out = b""
    for i in range(0, len(data), block):
        raw = data[i:i + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15); body = c.compress(raw) + c.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(raw) & 0xFFFFFFFF, len(raw))
    return out + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def test_blocked_gzip_input_is_read_as_byte_ranges(ds_small, tmp_path):
    """BGZF input (bgzip / this tool's -G output): the reader hops over the blocks by their size fields and -c threads inflate + parse byte ranges
    of the text. Same output as from the plain file, with ranges smaller than a block, of a few blocks, of one ticket; an ordinary .gz of the
    same text still goes through the one-thread reader; rtk_bgzip writes what python's gzip reads back."""
    import gzip
    text = open(ds_small + ".lr.fq", "rb").read()
    text = text[:text.index(b"\n@", 40000) + 1] if len(text) > 60000 else text
    plain, bg, gz = str(tmp_path / "in.fq"), str(tmp_path / "in.bgzf.fq.gz"), str(tmp_path / "in.plain.fq.gz")
    open(plain, "wb").write(text); open(bg, "wb").write(_bgzf_blocks(text)); gzip.open(gz, "wb").write(text)
    assert gzip.open(bg, "rb").read() == text
    exe = os.path.join(ROOT, "tests", "hostsim", "Ratatosk_sim")
    outs = []
    for tag, inp, batch in (("plain", plain, "3000"), ("bgzf_small_ranges", bg, "100"), ("bgzf_ticket_ranges", bg, "3000"), ("gz_stream", gz, "3000")):
        out = str(tmp_path / tag)
        r = subprocess.run([exe, "correct", "-1", "-c", "4", "-B", batch, "-g", ds_small + ".index.k31.fasta.gz", "-d", ds_small + ".index.k31.rtsk", "-l", inp, "-o", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(open(out + ".2.fastq", "rb").read())
    assert outs[0] == outs[1] == outs[2] == outs[3] and len(outs[0]) > 1000
    r = subprocess.run([exe, "correct", "-1", "--parse-only", "-c", "3", "-B", "100", "-l", bg], capture_output=True, text=True)
    assert r.returncode == 0 and "(1 plain or blocked gzip" in r.stdout and "%d reads" % (text.count(b"\n") // 4) in r.stdout, r.stdout + r.stderr
    r = subprocess.run([exe, "correct", "-1", "--parse-only", "-c", "3", "-B", "100", "-l", gz], capture_output=True, text=True)
    assert r.returncode == 0 and "(0 plain or blocked gzip" in r.stdout, r.stdout
    mine = str(tmp_path / "mine.gz")
    assert subprocess.run([os.path.join(BIN, "rtk_bgzip"), plain, mine, "-@", "3"]).returncode == 0
    assert gzip.open(mine, "rb").read() == text
    # a file that starts with a BGZF block and goes on as an ordinary member is not taken for blocked gzip
    mixed = str(tmp_path / "mixed.gz")
    open(mixed, "wb").write(_bgzf_blocks(text[:2000])[:-28] + gzip.compress(text[2000:]))
    r = subprocess.run([exe, "correct", "-1", "--parse-only", "-c", "3", "-l", mixed], capture_output=True, text=True)
    assert r.returncode == 0 and "(0 plain or blocked gzip" in r.stdout and "%d reads" % (text.count(b"\n") // 4) in r.stdout, r.stdout

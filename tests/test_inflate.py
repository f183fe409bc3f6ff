"""The host reader's own gzip decoder (ratatosk_amd/csrc/common/finflate.hpp inside common/mgzip.hpp), through `rtk_gunzip --check`, against
python's zlib: streams of every block kind and strategy, sizes from nothing to megabytes with matches at every distance up to 32 KB, files
of many members on several threads; and damaged input: every mutation must end in exit code 1 or in the right text, never in a crash or in
wrong text with exit code 0 (the member CRC sees to that)."""
import gzip
import os
import random
import subprocess
import zlib

from conftest import BIN

TOOL = os.path.join(BIN, "rtk_gunzip")


def _text(kind, n, rng):
    if kind == 0:
        return bytes(rng.getrandbits(8) for _ in range(n))
    if kind == 1:
        return bytes(rng.choice(b"ACGT") for _ in range(n))
    if kind == 2:
        return (b"ACGTTGCA" * (n // 8 + 1))[:n]
    if kind == 3:
        return b"A" * n
    if kind == 4:  # copies at every distance and length next to literals: what a FASTQ stream looks like to the decoder
        out = bytearray()
        while len(out) < n:
            if out and rng.random() < 0.5:
                d, l = rng.randint(1, min(len(out), 40000)), rng.randint(3, 258)
                for _ in range(l):
                    out.append(out[-d])
            else:
                out.append(rng.choice(b"ACGTNIIIHGF@+\n"))
        return bytes(out[:n])
    return bytes(rng.choice(b"AB") for _ in range(n))


def _check(path, threads="3"):
    r = subprocess.run([TOOL, path, "--check", "-@", threads], capture_output=True, text=True, timeout=120)
    return r.returncode, r.stdout.split(), r.stderr


def test_streams_of_every_kind_against_zlib(tmp_path):
    rng = random.Random(11)
    p = str(tmp_path / "c.gz")
    n_cases = 0
    for kind in range(6):
        for n in (0, 1, 7, 300, 5000, 70000, 400000):
            d = _text(kind, n, rng)
            for lvl in (0, 1, 6, 9):
                for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                    co = zlib.compressobj(lvl, zlib.DEFLATED, 31, 9, strat)
                    z = co.compress(d[:len(d) // 2]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(d[len(d) // 2:]) + co.flush()
                    open(p, "wb").write(z)
                    rc, out, err = _check(p, "1" if n_cases % 2 else "3")
                    assert rc == 0 and out[:2] == [str(len(d)), "%08x" % zlib.crc32(d)], (kind, n, lvl, strat, out, err)
                    n_cases += 1
    assert n_cases == 6 * 7 * 16


def test_many_members_and_the_text_itself(tmp_path):
    rng = random.Random(3)
    parts = [_text(4, rng.choice([0, 10, 90000, 700000]), rng) for _ in range(40)]
    p = str(tmp_path / "m.gz")
    open(p, "wb").write(b"".join(gzip.compress(x, rng.choice([0, 1, 6, 9])) for x in parts))
    text = b"".join(parts)
    for threads in ("1", "2", "8"):
        rc, out, err = _check(p, threads)
        assert rc == 0 and out == [str(len(text)), "%08x" % zlib.crc32(text), "40"], (threads, out, err)
    r = subprocess.run([TOOL, p, "-@", "5", "-o", str(tmp_path / "m.txt")], capture_output=True)
    assert r.returncode == 0 and open(str(tmp_path / "m.txt"), "rb").read() == text


def test_damaged_streams_are_errors_never_wrong_text(tmp_path):
    rng = random.Random(5)
    good = []
    for kind, lvl in ((1, 6), (4, 6), (4, 1), (4, 0), (0, 6), (3, 9), (5, 6)):
        d = _text(kind, rng.choice([3000, 60000, 200000]), rng)
        good.append((gzip.compress(d, lvl), d))
    p = str(tmp_path / "x.gz")
    n_err = n_ok = 0
    for it in range(400):
        z, d = rng.choice(good)
        b = bytearray(z)
        mode = rng.random()
        if mode < 0.5:
            for _ in range(rng.choice([1, 1, 2, 5])):
                b[rng.randrange(10, len(b))] = rng.getrandbits(8)
        elif mode < 0.7:
            b = b[:rng.randrange(10, len(b))]
        elif mode < 0.85:
            b = b[:10] + bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 3000)))
        else:
            at = rng.randrange(10, len(b))
            b[at:at] = bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 50)))
        open(p, "wb").write(bytes(b))
        rc, out, err = _check(p, rng.choice(["1", "3"]))
        assert rc in (0, 1), (it, rc, err)  # not a signal
        if rc == 0:
            assert out[:2] == [str(len(d)), "%08x" % zlib.crc32(d)], (it, out)  # the mutation hit a byte that does not matter (header time stamp, ...)
            n_ok += 1
        else:
            assert "damaged" in err
            n_err += 1
    assert n_err > 300

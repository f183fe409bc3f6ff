"""Bit-level identities that lane-parallel device code of round 4 rests on, restated in Python and checked against the plain rule they replace (the
1-lane host simulator runs the plain rules, so these identities have no other CPU-side check; the gpu tier holds the kernels to the oracle):
  * k_finalize's solid filter per 64-window block (csrc/hip/rtk_seeds.h: run starts OR-ed over a sliding window by doubling shifts) against the
    per-window rule "dropped iff the k - 1 presence bits behind the window have a one above their first zero" (src/Graph.cpp:221-239);
  * k_lookup_exact's k-mer codes from ballot bit planes (csrc/hip/rtk_device.hip) against the character-by-character 2-bit packing;
  * the 32-bit sort key of a weak hit (variant position, base mask, kind side by side) against the 64-bit key it stands for: same order, exact round trip;
  * the reverse complement four characters at a time (rtk_reverse_copy) against the character-by-character one."""
import random

M64 = (1 << 64) - 1


def _ref_keep(cur, nxt, k):
    kmask, keep = (1 << (k - 1)) - 1, 0
    for j in range(64):
        if (cur >> j) & 1:
            after = nxt if j == 63 else (((cur >> (j + 1)) | (nxt << (63 - j))) & M64)
            b = after & kmask
            if (b & (b + 1)) == 0:
                keep |= 1 << j
    return keep


def _shr128(lo, hi, s):
    return (((lo >> s) | (hi << (64 - s))) & M64, hi >> s) if s else (lo, hi)


def _fast_keep(cur, nxt, k):
    rl, rh = cur & ~(cur << 1) & M64, nxt & ~((nxt << 1) | (cur >> 63)) & M64
    w, have = k - 2, 1
    while 2 * have <= w:
        sl, sh = _shr128(rl, rh, have); rl |= sl; rh |= sh; have *= 2
    if have < w:
        sl, sh = _shr128(rl, rh, w - have); rl |= sl; rh |= sh
    return cur & ~_shr128(rl, rh, 2)[0] & M64


def test_solid_filter_by_blocks_is_the_per_window_rule():
    rnd = random.Random(1)
    for k in (3, 5, 15, 21, 25, 31, 33, 47, 63):
        for _ in range(3000):
            dens = rnd.choice([0.05, 0.3, 0.7, 0.95])
            cur = sum((rnd.random() < dens) << i for i in range(64))
            nxt = 0 if rnd.random() < 0.1 else sum((rnd.random() < dens) << i for i in range(64))
            assert _ref_keep(cur, nxt, k) == _fast_keep(cur, nxt, k), (k, hex(cur), hex(nxt))


def _spread(x):
    x = (x | (x << 16)) & 0x0000FFFF0000FFFF; x = (x | (x << 8)) & 0x00FF00FF00FF00FF; x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0F
    x = (x | (x << 2)) & 0x3333333333333333
    return (x | (x << 1)) & 0x5555555555555555


def _brev64(x):
    return int(format(x, "064b")[::-1], 2)


def test_kmer_codes_from_bit_planes():
    rnd = random.Random(2)
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    for k in (15, 21, 25, 31, 32):
        for _ in range(300):
            text = "".join(rnd.choice("ACGTACGTACGTN") for _ in range(128))
            ch = [ord(c) for c in text]
            planes = lambda f, half: sum(int(f(ch[64 * half + i])) << i for i in range(64))
            lo = [planes(lambda c: ((c >> 1) ^ (c >> 2)) & 1, h) for h in (0, 1)]; hi = [planes(lambda c: (c >> 2) & 1, h) for h in (0, 1)]
            va = [planes(lambda c: chr(c) in "ACGT", h) for h in (0, 1)]
            km = (1 << k) - 1
            for lane in range(64):
                take = lambda p: ((p[0] >> lane) | (p[1] << (64 - lane)) if lane else p[0]) & km
                p0, p1, pv = take(lo), take(hi), take(va)
                window = text[lane:lane + k]
                ok = all(c in code for c in window)
                assert (pv == km) == ok
                if ok:
                    want = 0
                    for c in window:
                        want = (want << 2) | code[c]
                    got = (_spread(_brev64(p1) >> (64 - k)) << 1) | _spread(_brev64(p0) >> (64 - k))
                    assert got == want, (k, lane, window)


def test_weak_hit_sort_key_in_32_bits():
    rnd = random.Random(3)
    keys = []
    for _ in range(5000):
        pos, kind = rnd.randrange(0, 1 << 20), rnd.choice((1, 2, 3))
        mis = 0 if kind == 3 and rnd.random() < 0.5 else 1 << rnd.randrange(4)
        keys.append((pos << 16) | (mis << 8) | kind)
    k32 = lambda key: ((key >> 16) << 6) | (((key >> 8) & 15) << 2) | (key & 3)
    back = lambda K: ((K >> 6) << 16) | (((K >> 2) & 15) << 8) | (K & 3)
    assert all(back(k32(x)) == x and k32(x) < (1 << 32) for x in keys)
    assert sorted(range(len(keys)), key=lambda i: (keys[i], i)) == sorted(range(len(keys)), key=lambda i: (k32(keys[i]), i))


def test_reverse_complement_by_words():
    rnd = random.Random(4)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "M": "K", "K": "M", "R": "Y", "Y": "R", "V": "B", "B": "V", "H": "D", "D": "H"}
    tab = [ord(comp.get(chr(c), chr(c))) for c in range(256)]
    for n in list(range(0, 12)) + [63, 64, 65, 255, 1000, 1001, 1002, 1003]:
        src = [ord(rnd.choice("ACGTNMKRYVBHDacgt")) for _ in range(n)]
        dst = [None] * n
        n4 = n & ~3
        for i in range(0, n4, 4):  # one lane's access: the word at src[n - 4 - i], its last character first
            w = src[n - 4 - i:n - i]
            dst[i:i + 4] = [tab[w[3]], tab[w[2]], tab[w[1]], tab[w[0]]]
        for i in range(n4, n):
            dst[i] = tab[src[n - 1 - i]]
        assert dst == [tab[c] for c in reversed(src)]

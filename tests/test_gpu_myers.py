"""HIP Myers kernels (wave-parallel, one query word per lane) through the C ABI against the reference golden vectors
and the oracle on seeded random problems, including row-blocked (>64 words) and Hirschberg-sized ones. Bit-exact."""
import random

import pytest

from conftest import golden_rows
from oracle import oracle_py as op
from ratatosk_amd import api

pytestmark = pytest.mark.gpu


def test_gpu_myers_golden_distance_and_locations():
    rows = [r for r in golden_rows() if not r["path"]]
    res = api.myers_batch([r["q"] for r in rows], [r["t"] for r in rows], [r["k"] for r in rows], [r["mode"] for r in rows], want_path=False)
    for r, (d, locs, _) in zip(rows, res):
        assert d == r["d"], (len(r["q"]), len(r["t"]), r["mode"], r["k"])
        assert locs == r["locs"]


def test_gpu_myers_golden_paths():
    rows = [r for r in golden_rows() if r["path"]]
    res = api.myers_batch([r["q"] for r in rows], [r["t"] for r in rows], [r["k"] for r in rows], [r["mode"] for r in rows], want_path=True)
    for r, (d, locs, cig) in zip(rows, res):
        assert d == r["d"] and locs == r["locs"] and cig == r["cigar"], (len(r["q"]), len(r["t"]), r["mode"])


def test_gpu_myers_random_vs_oracle_large():
    rnd = random.Random(17)
    Q, T, M = [], [], []
    for m, n in [(4200, 900), (9000, 400), (20000, 1000), (700, 30000), (5000, 5000), (64 * 64, 300), (64 * 64 + 1, 300), (64 * 128, 100)]:
        q = "".join(rnd.choice("ACGT") for _ in range(m))
        base = q if n <= m else q + "".join(rnd.choice("ACGT") for _ in range(n - m))
        t = "".join(c if rnd.random() > 0.08 else rnd.choice("ACGT") for c in base[:n])
        for mode in (0, 1):
            Q.append(q); T.append(t); M.append(mode)
    res = api.myers_batch(Q, T, [-1] * len(Q), M, want_path=True)
    for q, t, m, r in zip(Q, T, M, res):
        o = op.myers(q, t, -1, m, True)
        assert (r[0], r[1], r[2]) == (o[0], o[1], o[2]), (len(q), len(t), m)
    res = api.myers_batch(Q, T, [-1] * len(Q), [2] * len(Q), want_path=False)
    for q, t, r in zip(Q, T, res):
        o = op.myers(q, t, -1, 2, False)
        assert (r[0], r[1]) == (o[0], o[1])

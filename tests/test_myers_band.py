"""Ukkonen band of the big NW problems (reference: src/edlib.cpp:194-212 k doubling, :744-775 and :778-915 banded block ranges).
The device computes only the band of k (ring schedule on one wave, banded row blocks on several: csrc/hip/rtk_myers.h); the oracle computes
whole columns. Bounded distances with k = exact / exact - 1 / 0 / generous, unknown distances whose first guess is too small, and
Hirschberg-sized paths must be the oracle's, bit for bit. CPU tier: the simulator (band logic, column extraction, split search);
GPU tier: the wave schedules themselves, single wave and multi-wave workgroups."""
import random

import pytest

from conftest import SIM_LIB
from oracle import oracle_py as op
from ratatosk_amd import api


def _mutate(rnd, q, err):
    out = []
    for c in q:
        r = rnd.random()
        if r < err / 3:
            continue
        if r < 2 * err / 3:
            out.append(rnd.choice("ACGT")); out.append(c)
        elif r < err:
            out.append(rnd.choice("ACGT"))
        else:
            out.append(c)
    return "".join(out)


def _problems(seed, cases, extra=True):
    rnd = random.Random(seed)
    Q, T = [], []
    for m, e in cases:
        q = "".join(rnd.choice("ACGT") for _ in range(m))
        Q.append(q); T.append(_mutate(rnd, q, e))
    if extra:
        Q.append("".join(rnd.choice("ACGT") for _ in range(6000))); T.append("".join(rnd.choice("ACGT") for _ in range(1500)))  # |n - m| = 4500: row blocks, band wider than the ring
        Q.append("".join(rnd.choice("ACGT") for _ in range(1500))); T.append("".join(rnd.choice("ACGT") for _ in range(7000)))  # one row block, long target
        Q.append("".join(rnd.choice("ACGT") for _ in range(9000))); T.append("".join(rnd.choice("ACGT") for _ in range(9100)))  # unrelated: the first guess of the distance is too small
        q = "".join(rnd.choice("ACGTN") for _ in range(5000)); Q.append(q); T.append(_mutate(rnd, q, 0.1).replace("G", "R", 3).replace("A", "N", 2))  # IUPAC / N on both sides
    return Q, T


def _check(Q, T, lib_path, waves=0):
    exact = [op.myers(q, t, -1, 0, False)[0] for q, t in zip(Q, T)]
    qq, tt, kk = [], [], []
    for q, t, d in zip(Q, T, exact):
        for k in (-1, d, d - 1, 0, d + 7, 10 * d + 1):
            qq.append(q); tt.append(t); kk.append(k)
    res = api.myers_batch(qq, tt, kk, [0] * len(qq), want_path=False, lib_path=lib_path, waves=waves)
    for q, t, k, r in zip(qq, tt, kk, res):
        o = op.myers(q, t, k, 0, False)
        assert (r[0], r[1]) == (o[0], o[1]), (len(q), len(t), k)
    res = api.myers_batch(Q, T, [-1] * len(Q), [0] * len(Q), want_path=True, lib_path=lib_path, waves=waves)
    for q, t, r in zip(Q, T, res):
        o = op.myers(q, t, -1, 0, True)
        assert (r[0], r[1], r[2]) == (o[0], o[1], o[2]), (len(q), len(t))


CASES_SMALL = [(4200, 0.1), (9000, 0.08), (5000, 0.3), (12000, 0.02), (4097, 0.5), (8200, 0.0), (20000, 0.1)]


def test_sim_banded_nw_against_oracle():
    Q, T = _problems(5, CASES_SMALL)
    _check(Q, T, SIM_LIB)


def test_band_geometry_of_the_header():
    """The band the device uses: |d| + |(n - m) - d| <= k, symmetric under d -> (n - m) - d (the reversed half passes use the same numbers)."""
    import ctypes
    for m, n, k in [(100, 100, 10), (100, 130, 40), (130, 100, 40), (50, 60, 5), (60, 50, 10), (7, 7, 0)]:
        d, ad = n - m, abs(n - m)
        h = (k - ad) // 2 if k > ad else 0
        lo, hi = min(0, d) - h, max(0, d) + h
        inside = [x for x in range(-m, n + 1) if abs(x) + abs(d - x) <= max(k, ad)]
        assert (lo, hi) == (min(inside), max(inside))
        assert (d - hi, d - lo) == (lo, hi)


@pytest.mark.gpu
def test_gpu_banded_nw_single_wave():
    Q, T = _problems(7, CASES_SMALL + [(40000, 0.1), (70000, 0.12)])
    _check(Q, T, None)


@pytest.mark.gpu
def test_gpu_banded_nw_multi_wave():
    Q, T = _problems(9, CASES_SMALL + [(40000, 0.1), (98000, 0.1)])
    _check(Q, T, None, waves=16)
    _check(Q[:4], T[:4], None, waves=8)
